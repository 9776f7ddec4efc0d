#!/usr/bin/env python3
"""Monte-Carlo loop rate (noise generation + decode + classification all inside the kernel): frames/s per SNR."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acg_alp_ldpc_amd as A
H = A.read_pcm("data/H05.txt")
G, _ = H.get_orthogonal()
cws = A.gen_random_codewords(G, 8192, 239239239)
F = 1 << 22
for name, dec in (("BP-50", A.BeliefPropagationDecoder(50)), ("MS-50", A.MinSumDecoder(50, 0.75)), ("QP-ADMM-100", A.QPADMMDecoder(1.95, 0.5, 100, 1e-5))):
    for snr in (-3.0, -2.0, 0.0, 2.0):
        f = F if name != "QP-ADMM-100" else F // 8
        A.run_experiment(dec, cws, H, snr, frames=4096, noise="device", seed=1)
        r = A.run_experiment(dec, cws, H, snr, frames=f, noise="device", seed=1)
        print("%-12s %+.1f dB: %8.2f M frames/s (kernel %.1f ms, wall %.3f s) FER %.5f mean iters %.2f pseudo %d"
              % (name, snr, f / (r.kernel_ms * 1e-3) / 1e6, r.kernel_ms, r.time_sec, r.FER(), r.mean_iters(), r.pseudo))
