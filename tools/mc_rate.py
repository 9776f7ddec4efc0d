#!/usr/bin/env python3
"""Monte-Carlo loop rate (noise generation + decode + classification all on the device): frames/s per SNR.

    python tools/mc_rate.py [--algo bp,minsum,qpadmm] [--matrix data/H05.txt] [--snrs -3,-2,0,2]
                            [--alpha 1.95 --mu 0.5] [--frames N]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acg_alp_ldpc_amd as A

ap = argparse.ArgumentParser()
ap.add_argument("--algo", default="bp,minsum,qpadmm")
ap.add_argument("--matrix", default="data/H05.txt")
ap.add_argument("--snrs", default="-3,-2,0,2")
ap.add_argument("--alpha", type=float, default=1.95)
ap.add_argument("--mu", type=float, default=0.5)
ap.add_argument("--frames", type=int, default=1 << 22)
a = ap.parse_args()

H = A.read_pcm(a.matrix)
G, _ = H.get_orthogonal()
cws = A.gen_random_codewords(G, 8192, 239239239)
decs = {"bp": ("BP-50", lambda: A.BeliefPropagationDecoder(50)), "minsum": ("MS-50", lambda: A.MinSumDecoder(50, 0.75)),
        "qpadmm": ("QP-ADMM-100", lambda: A.QPADMMDecoder(a.alpha, a.mu, 100, 1e-5)),
        # layered schedule (FER-level parity only): half the iterations
        "bp_layered": ("BP-25-layered", lambda: A.BeliefPropagationDecoder(25, schedule=A.SCHEDULE_LAYERED)),
        "bp_layered_f16": ("BP-25-lay-f16", lambda: A.BeliefPropagationDecoder(25, schedule=A.SCHEDULE_LAYERED, precision=A.PREC_F16)),
        "minsum_layered": ("MS-25-layered", lambda: A.MinSumDecoder(25, 0.75, schedule=A.SCHEDULE_LAYERED)),
        "minsum_layered_f16": ("MS-25-lay-f16", lambda: A.MinSumDecoder(25, 0.75, schedule=A.SCHEDULE_LAYERED, precision=A.PREC_F16))}
for key in a.algo.split(","):
    name, make = decs[key]
    dec = make()
    for snr in [float(x) for x in a.snrs.split(",")]:
        f = a.frames if key != "qpadmm" else a.frames // 8
        A.run_experiment(dec, cws, H, snr, frames=4096, noise="device", seed=1)
        r = A.run_experiment(dec, cws, H, snr, frames=f, noise="device", seed=1)
        print("%-14s %+.1f dB: %8.2f M frames/s by wall time (%.3f s; decode kernel alone %.1f ms) FER %.6f mean iters %.2f pseudo %d"
              % (name, snr, f / r.time_sec / 1e6, r.time_sec, r.kernel_ms, r.FER(), r.mean_iters(), r.pseudo), flush=True)
