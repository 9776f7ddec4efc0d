"""TEST INFRASTRUCTURE — generates tests/golden/* from the REAL reference.

Run in the build container only (needs /root/reference to build oracle/_ref/libacg_ref.so):

    python oracle/make_golden.py

Everything written is data (inputs + the reference's outputs); no reference source text.
Frame f (0-based) of every set uses the reference's own generator chain:
    G = GetOrtogonal(H)                       utils/codeword.h:97
    codewords = gen_random_codewords(G, mt19937(239'239'239))   main.cpp:63-64
    y_f = transmit(snr, codeword_f, mt19937(f+1))               experiment.h:97-99
The reference's global BP node counter is reset before every decode so a fixture does not
depend on how many frames were decoded before it (it only changes a summation order).
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.pyoracle import Ref  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")

MATRICES = {
    # name: (file, (alpha, mu) used by the reference for this matrix or a valid pair)
    "H": ("H.txt", (1.95, 0.5)),          # no reference pair for H.txt; e_min=8 so the guard passes
    "H05": ("H05.txt", (1.95, 0.5)),       # main.cpp:33
    "optimalH": ("optimalH.txt", (1.2, 0.55)),  # main.cpp:31
}
SNRS = [-3.0, -2.0, 0.0, 2.0]
FRAMES = 64
BP_ITERS = [1, 2, 5, 20, 50]
ADMM_ITERS = [1, 2, 5, 10, 100]
TRACE_FRAMES = 2
TRACE_ITERS = [0, 1, 2]
TRACE_SNRS = [-2.0, 2.0]


def main():
    os.makedirs(OUT, exist_ok=True)
    r = Ref()
    known = {"sigma_pins": {}, "experiments": [], "structure": {}}

    for name, (fn, (alpha, mu)) in MATRICES.items():
        H = r.read_pcm(os.path.join(ROOT, "data", fn))
        G, ok = r.get_orthogonal(H)
        assert ok
        cws = r.gen_codewords(G, 239239239, 1000)
        known["structure"][name] = dict(m=int(H.shape[0]), n=int(H.shape[1]), E=int(H.sum()),
                                        k=int(G.shape[0]), admm=r.admm_shape(H),
                                        cw0_first32="".join(map(str, cws[0][:32])))
        for snr in SNRS:
            d = {}
            y = r.transmit_frames(cws[:FRAMES], snr)
            d["snr"] = np.float64(snr)
            d["cw"] = np.packbits(cws[:FRAMES], axis=1)
            d["y"] = y
            for it in BP_ITERS:
                bits = np.zeros((FRAMES, H.shape[1]), np.uint8)
                okf = np.zeros(FRAMES, np.uint8)
                for f in range(FRAMES):
                    r.reset_node_counter()
                    b, o, _ = r.bp_decode(H, y[f], snr, it)
                    bits[f], okf[f] = b[0], o[0]
                d["bp%d_bits" % it] = np.packbits(bits, axis=1)
                d["bp%d_ok" % it] = okf
            d["admm_alpha_mu"] = np.array([alpha, mu])
            for it in ADMM_ITERS:
                for tag, eps in (("e0", 0.0), ("e5", 1e-5)):
                    b, o, _ = r.qpadmm_decode(H, y, snr, alpha, mu, it, eps)
                    d["admm%d_%s_bits" % (it, tag)] = np.packbits(b, axis=1)
                    d["admm%d_%s_ok" % (it, tag)] = o
            if snr in TRACE_SNRS:
                for it in TRACE_ITERS:
                    tr = [None] * TRACE_FRAMES
                    for f in range(TRACE_FRAMES):
                        r.reset_node_counter()
                        tr[f] = r.bp_trace(H, y[f], snr, it)
                    for k in ("c2v", "v2c_mag", "v2c_sgn", "post"):
                        d["trace%d_%s" % (it, k)] = np.stack([t[k] for t in tr])
            path = os.path.join(OUT, "%s_snr%+.0f.npz" % (name, snr))
            np.savez_compressed(path, **d)
            print("wrote", path, os.path.getsize(path) // 1024, "KiB")

    # known answers: the reference's experiment.h loop, ONE thread, 1000 frames
    H05 = r.read_pcm(os.path.join(ROOT, "data", "H05.txt"))
    Hopt = r.read_pcm(os.path.join(ROOT, "data", "optimalH.txt"))
    Hs = r.read_pcm(os.path.join(ROOT, "data", "H.txt"))
    G05file = r.read_pcm(os.path.join(ROOT, "data", "G05.txt"))
    runs = [
        ("H", Hs, "bp", 20, 0, 0, [2.0]),
        ("H05", H05, "bp", 50, 0, 0, [-3.0, -2.0, -1.0, 0.0]),
        ("H05", H05, "qpadmm", 100, 1.95, 0.5, [-3.0, -2.0, -1.0, 0.0]),
        ("optimalH", Hopt, "qpadmm", 100, 1.2, 0.55, [-3.0, -2.0]),
        ("optimalH", Hopt, "bp", 50, 0, 0, [-2.0]),
    ]
    for name, H, kind, it, alpha, mu, snrs in runs:
        G, _ = r.get_orthogonal(H)
        cws = r.gen_codewords(G, 239239239, 1000)
        for snr in snrs:
            r.reset_node_counter()
            res = r.experiment(kind, H, cws, snr, it, alpha, mu, 1e-5)
            res.pop("time_sec")
            res.update(matrix=name, kind=kind, max_iter=it, alpha=alpha, mu=mu, snr=snr, frames=1000,
                       codewords="GetOrtogonal+mt19937(239239239)")
            known["experiments"].append(res)
            print(res)
    # the reference's non-OPTIMAL path: codewords from data/G05.txt (main.cpp:59-60)
    cws = r.gen_codewords(G05file, 239239239, 1000)
    known["structure"]["H05"]["cw0_first32_G05file"] = "".join(map(str, cws[0][:32]))
    res = r.experiment("qpadmm", H05, cws, -2.0, 100, 1.95, 0.5, 1e-5)
    res.pop("time_sec")
    res.update(matrix="H05", kind="qpadmm", max_iter=100, alpha=1.95, mu=0.5, snr=-2.0, frames=1000,
               codewords="G05.txt+mt19937(239239239)")
    known["experiments"].append(res)

    for snr in [-5, -4.5, -4, -3.5, -3, -2.5, -2, -1.5, -1, -0.5, 0.0]:
        known["sigma_pins"]["%g" % snr] = float(np.sqrt(r.llr_variance(snr)))
    with open(os.path.join(OUT, "known_answers.json"), "w") as f:
        json.dump(known, f, indent=1, sort_keys=True)
    print("wrote known_answers.json")


if __name__ == "__main__":
    main()
