#!/usr/bin/env python3
"""Developer tool (GPU box): rate of ONE decoder configuration on a resident batch — for A/B runs of kernel variants.

    python tools/rate.py --algo qpadmm --iters 100 --frames 262144            # fixed work (eps_stop 0)
    python tools/rate.py --algo bp --engine streamed --frames 1048576
    python tools/rate.py --algo minsum --synthetic 5000 10000 3 6 --frames 32768 --snr 2
Prints one line: rate, mean kernel ms (HIP events on the launch stream), FER, mean sweeps, layout.
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", choices=["bp", "minsum", "qpadmm"], default="bp")
    ap.add_argument("--engine", choices=["auto", "fused", "streamed"], default="auto")
    ap.add_argument("--frames", type=int, default=1 << 20)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--snr", type=float, default=-2.0)
    ap.add_argument("--exit", action="store_true", help="the reference's stopping rule instead of fixed work")
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--prec", choices=["default", "f32", "f64", "f16"], default="default")
    ap.add_argument("--alpha", type=float, default=1.95)
    ap.add_argument("--mu", type=float, default=0.5)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--matrix", default=os.path.join(ROOT, "data", "H05.txt"))
    ap.add_argument("--synthetic", type=int, nargs=4, metavar=("M", "N", "DV", "DC"), default=None)
    ap.add_argument("--qc", type=int, nargs=3, metavar=("MB", "NB", "Z"), default=None,
                    help="all-ones MB x NB protograph of Z x Z cyclic shifts (shift = (3r + 7c) mod Z): a structured code whose variable sweep "
                         "walks the message lines in a few sequential streams")
    ap.add_argument("--layered", action="store_true", help="min-sum with the layered schedule (SCHEDULE_LAYERED)")
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    import numpy as np
    import torch
    import acg_alp_ldpc_amd as A
    from acg_alp_ldpc_amd._lib import McCfg, check, lib
    if a.qc:
        mb, nb, Z = a.qc
        Hd = np.zeros((mb * Z, nb * Z), dtype=np.uint8)
        k = np.arange(Z)
        for r in range(mb):
            for c in range(nb):
                Hd[r * Z + k, c * Z + (k + 3 * r + 7 * c) % Z] = 1
        H = A.ParityCheckMatrix(Hd)
        cws = np.zeros((1, nb * Z), dtype=np.uint8)
    elif a.synthetic:
        m, n, dv, dc = a.synthetic
        H = A.ParityCheckMatrix(A.regular_ldpc(m, n, dv, dc, seed=1))
        cws = np.zeros((1, n), dtype=np.uint8)
    else:
        H = A.read_pcm(a.matrix)
        G, ok = H.get_orthogonal()
        cws = A.gen_random_codewords(G, 8192, 239239239)
    eng = {"auto": A.ENGINE_AUTO, "fused": A.ENGINE_FUSED, "streamed": A.ENGINE_STREAMED}[a.engine]
    prec = {"default": A.PREC_DEFAULT, "f32": A.PREC_F32, "f64": A.PREC_F64, "f16": A.PREC_F16}[a.prec]
    t0 = time.time()
    if a.algo == "qpadmm":
        dec = A.QPADMMDecoder(a.alpha, a.mu, a.iters, 1e-5 if a.exit else 0.0, lanes_per_frame=a.lanes, precision=prec)
    elif a.algo == "minsum":
        dec = A.MinSumDecoder(a.iters, 0.75, early_exit=a.exit, lanes_per_frame=a.lanes, engine=eng, precision=prec,
                              schedule=A.SCHEDULE_LAYERED if a.layered else A.SCHEDULE_FLOODING)
    else:
        dec = A.BeliefPropagationDecoder(a.iters, early_exit=a.exit, lanes_per_frame=a.lanes, engine=eng, precision=prec,
                                         schedule=A.SCHEDULE_LAYERED if a.layered else A.SCHEDULE_FLOODING)
    h, _ = dec.handle(H)
    t_create = time.time() - t0
    F, n, nw = a.frames, H.n, (H.n + 31) // 32
    y = torch.empty((F, n), dtype=torch.float32, device="cuda")
    bits = torch.zeros((F, nw), dtype=torch.int32, device="cuda")
    okf = torch.zeros(F, dtype=torch.uint8, device="cuda")
    its = torch.zeros(F, dtype=torch.int32, device="cuda")
    stream = torch.cuda.Stream()
    cfg = McCfg()
    cfg.frames, cfg.first_frame, cfg.snr, cfg.seed, cfg.noise = F, 0, a.snr, 1, 0
    cfg.codewords, cfg.n_codewords = cws.ctypes.data, cws.shape[0]
    check(lib().acg_ldpc_awgn_dev(h, C.byref(cfg), y.data_ptr(), stream.cuda_stream))
    torch.cuda.synchronize()

    def step():
        dec.decode_batch_dev(H, y.data_ptr(), False, F, a.snr, bits.data_ptr(), okf.data_ptr(), its.data_ptr(), stream.cuda_stream)
    step()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    t0 = time.perf_counter()
    for e0, e1 in ev:
        e0.record(stream)
        step()
        e1.record(stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms = sum(e0.elapsed_time(e1) for e0, e1 in ev) / a.steps
    pad = np.zeros((cws.shape[0], nw * 32), dtype=np.uint8)
    pad[:, :n] = cws
    cwp = torch.from_numpy(np.packbits(pad, axis=1, bitorder="little").view(np.int32).copy()).cuda()
    idx = torch.arange(F, device="cuda") % cwp.shape[0]
    good = ((bits == cwp[idx]).all(dim=1) & (okf == 1)).sum().item()
    print("%s %s%s: %.3f M frames/s  kernel %.3f ms  fer %.5f  mean sweeps %.2f  create %.2f s  layout %s"
          % (a.tag, a.algo, " exit" if a.exit else " fixed", F * a.steps / dt / 1e6, kms, 1 - good / F, its.double().mean().item(),
             t_create, dec.layout(H)))


if __name__ == "__main__":
    main()
