#!/usr/bin/env python3
"""Soak check (GPU): the same device-noise Monte-Carlo run through every engine must give identical counters.

    python tools/soak_engines.py [--frames 8388608]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acg_alp_ldpc_amd as A

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1 << 23)
a = ap.parse_args()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = A.read_pcm(os.path.join(ROOT, "data", "H05.txt"))
G, _ = H.get_orthogonal()
cws = A.gen_random_codewords(G, 8192, 239239239)


def counters(r):
    return (r.correct, r.pseudo, r.total, r.sum_hamming, r.sum_hamming_ok, r.sum_hamming_wrong, round(r.mean_iters() * r.total))


bad = 0
for snr in (-2.0, 0.0):
    ref = None
    for tag, mk in (("fused L=32", lambda: A.BeliefPropagationDecoder(50, lanes_per_frame=32)),
                    ("fused L=64", lambda: A.BeliefPropagationDecoder(50, lanes_per_frame=64)),
                    ("fused L=16", lambda: A.BeliefPropagationDecoder(50, lanes_per_frame=16)),
                    ("workgroup L=256", lambda: A.BeliefPropagationDecoder(50, lanes_per_frame=256)),
                    ("streamed (ring)", lambda: A.BeliefPropagationDecoder(50, engine=A.ENGINE_STREAMED)),
                    ("streamed (VGPR)", lambda: (os.environ.__setitem__("ACG_STREAM_NO_RING", "1"), A.BeliefPropagationDecoder(50, engine=A.ENGINE_STREAMED),
                                                   )[1])):
        dec = mk()
        dec.handle(H)
        os.environ.pop("ACG_STREAM_NO_RING", None)
        r = A.run_experiment(dec, cws, H, snr, frames=a.frames, noise="device", seed=7)
        dec.close()
        c = counters(r)
        if ref is None:
            ref = c
        ok = c == ref
        bad += not ok
        print("BP-50 %+.1f dB %-16s %s %s" % (snr, tag, c, "ok" if ok else "MISMATCH"), flush=True)
    ref = None
    for tag, mk in (("workgroup (QC tuples)", lambda: A.QPADMMDecoder(1.95, 0.5, 100, 1e-5)),
                    ("workgroup (annealed)", lambda: (os.environ.__setitem__("ACG_ADMM_NO_QC", "1"), A.QPADMMDecoder(1.95, 0.5, 100, 1e-5))[1]),
                    ("wave L=64", lambda: A.QPADMMDecoder(1.95, 0.5, 100, 1e-5, lanes_per_frame=64)),
                    ("wave L=32", lambda: A.QPADMMDecoder(1.95, 0.5, 100, 1e-5, lanes_per_frame=32))):
        dec = mk()
        dec.handle(H)
        os.environ.pop("ACG_ADMM_NO_QC", None)
        r = A.run_experiment(dec, cws, H, snr, frames=a.frames // 8, noise="device", seed=7)
        dec.close()
        c = counters(r)
        if ref is None:
            ref = c
        ok = c == ref
        bad += not ok
        print("QP-ADMM-100 %+.1f dB %-22s %s %s" % (snr, tag, c, "ok" if ok else "MISMATCH"), flush=True)
sys.exit(1 if bad else 0)
