#!/usr/bin/env python3
"""Soak of the layered min-sum kernel against its numpy restatement (tests/layered_ref.py) at volume: every word, flag and
iteration count of N frames, fp32 and fp16 message storage, early exit and fixed work.  The restatement is the repo's own
(parity unpinned): this checks that the kernel is deterministic and does what its description says on rare paths too
(frames that run out of iterations and pass the explicit syndrome, ties between the two minima, zero posteriors)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import acg_alp_ldpc_amd as A
    from layered_ref import layered_minsum
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    for name, snr in (("H05.txt", -2.0), ("optimalH.txt", -1.0)):
        H = A.read_pcm(os.path.join(ROOT, "data", name))
        Hm = H.dense()
        G, _ = H.get_orthogonal()
        cws = A.gen_random_codewords(G, 4096, 239239239)
        _, _, layers = H.layers()
        y = A.transmit_frames(cws, snr, first_frame=0, frames=N)
        for prec, dt in ((A.PREC_DEFAULT, np.float32), (A.PREC_F16, np.float16)):
            t0 = time.time()
            rb, rok, rit = layered_minsum(Hm, layers, y, snr, 25, 0.75, dt)
            t_ref = time.time() - t0
            for ee in (True, False):
                dec = A.MinSumDecoder(25, 0.75, schedule=A.SCHEDULE_LAYERED, early_exit=ee, precision=prec)
                bits, ok, iters = dec.decode_batch(H, y, snr)
                dec.close()
                dw = int((bits != rb).any(axis=1).sum())
                print("%-12s %+.1f dB %s messages, %s: %d frames, differing words %d, flags %d, iteration counts %d; ok %.4f, out of iterations but "
                      "codeword %d (restatement %.0f s)" % (name, snr, "fp16" if dt == np.float16 else "fp32", "early exit" if ee else "fixed work", N, dw,
                                                            int((ok != rok).sum()), int((iters != rit).sum()), ok.mean(),
                                                            int(((rok == 1) & (rit == 25)).sum()), t_ref), flush=True)


if __name__ == "__main__":
    main()
