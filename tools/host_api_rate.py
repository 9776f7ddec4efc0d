#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point acg_ldpc_decode_batch (doubles in, bytes out): DESIGN.md §5."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import acg_alp_ldpc_amd as A
H = A.read_pcm("data/H05.txt")
G, _ = H.get_orthogonal()
cws = A.gen_random_codewords(G, 4096, 1)
F = 1 << 19
rng = np.random.default_rng(0)
snr = -2.0
y = (1 - 2 * cws[np.arange(F) % 4096].astype(np.float64)) + np.sqrt(A.llr_variance(snr)) * rng.standard_normal((F, H.n))
dec = A.BeliefPropagationDecoder(50)
dec.decode_batch(H, y[:1024], snr)
for _ in range(2):
    t = time.time(); bits, ok, it = dec.decode_batch(H, y, snr); dt = time.time() - t
    print("host API: %d frames in %.3f s = %.2f M frames/s (H2D %.0f MB, kernel %.1f ms), ok=%.4f" % (F, dt, F / dt / 1e6, y.nbytes / 1e6, dec.last_kernel_ms(H), ok.mean()))
