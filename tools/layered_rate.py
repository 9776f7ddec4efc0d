#!/usr/bin/env python3
"""frames/s and FER of the layered min-sum schedule next to flooding min-sum and sum-product (device-resident batch)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import acg_alp_ldpc_amd as A
    import bench
    a = type("a", (), dict(inproc=0, gpus=1))()
    rig = bench.Rig(a)
    for name in ("H05.txt", "optimalH.txt"):
        H = A.read_pcm(os.path.join(bench.ROOT, "data", name))
        G, _ = H.get_orthogonal()
        cws = A.gen_random_codewords(G, 8192, 239239239)
        F = 1 << 20
        batch = bench.Batch(rig, H, cws, F)
        for snr in (-2.0, 2.0):
            for tag, ctor in (("flooding ms 50 fixed", lambda dev: A.MinSumDecoder(50, 0.75, early_exit=False, device=dev)),
                              ("layered  ms 25 fixed", lambda dev: A.MinSumDecoder(25, 0.75, early_exit=False, device=dev, schedule=A.SCHEDULE_LAYERED)),
                              ("flooding ms 50 exit ", lambda dev: A.MinSumDecoder(50, 0.75, early_exit=True, device=dev)),
                              ("layered  ms 25 exit ", lambda dev: A.MinSumDecoder(25, 0.75, early_exit=True, device=dev, schedule=A.SCHEDULE_LAYERED)),
                              ("sum-product 50 exit ", lambda dev: A.BeliefPropagationDecoder(50, early_exit=True, device=dev))):
                r = bench.decode_leg(rig, batch, ctor, snr, 3, 1)
                print("%-12s %+.1f dB  %s  %8.2f M frames/s  kernel %7.2f ms  FER %.5f  mean iters %.2f  [%s]"
                      % (name, snr, tag, r["value"] / 1e6, r["kernel_ms"], r["fer"], r["mean_iters"], r["instance"][:110]), flush=True)
        del batch


if __name__ == "__main__":
    main()
