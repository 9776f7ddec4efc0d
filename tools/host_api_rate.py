#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer entry points: acg_ldpc_decode_batch (doubles in, one byte per
bit out), acg_ldpc_decode_batch_f32 (floats in), and the latency of the reference-shaped single-frame decode()."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import acg_alp_ldpc_amd as A

H = A.read_pcm(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "H05.txt"))
G, _ = H.get_orthogonal()
cws = A.gen_random_codewords(G, 4096, 1)
F = 1 << 20
rng = np.random.default_rng(0)
snr = -2.0
y = (1 - 2 * cws[np.arange(F) % 4096].astype(np.float64)) + np.sqrt(A.llr_variance(snr)) * rng.standard_normal((F, H.n))
y32 = y.astype(np.float32)
for name, dec in (("BP-50", A.BeliefPropagationDecoder(50)), ("QP-ADMM-100", A.QPADMMDecoder(1.95, 0.5, 100, 1e-5))):
    dec.decode_batch(H, y[:1024], snr)
    out = (np.zeros((F, H.n), np.uint8), np.zeros(F, np.uint8), np.zeros(F, np.int32))   # touched once: no page faults in the timed calls
    for arr, what in ((y, "float64 symbols (exact LLRs)"), (y32, "float32 symbols")):
        best = 1e9
        for _ in range(3):
            t = time.time()
            bits, ok, it = dec.decode_batch(H, arr, snr, out=out)
            best = min(best, time.time() - t)
        print("%s host API, %s: %d frames in %.3f s = %.2f M frames/s (H2D %.0f MB, D2H-side %.0f MB of bytes), ok=%.4f"
              % (name, what, F, best, F / best / 1e6, arr.nbytes / 1e6, bits.nbytes / 1e6, ok.mean()))
    ts = []
    for f in range(200):
        t = time.perf_counter()
        dec.decode(H, y[f], snr)
        ts.append(time.perf_counter() - t)
    ts = np.array(ts[20:]) * 1e6
    print("%s single-frame decode(H, y, snr): median %.0f us, p90 %.0f us" % (name, np.median(ts), np.percentile(ts, 90)))
