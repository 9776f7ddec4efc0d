"""CPU: oracle restatement vs the real reference compiled as-is (oracle/_ref).  Skipped when the
reference tree / prebuilt .so is absent."""
import os

import numpy as np
import pytest

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")


def test_parse_and_generators(oracle, ref):
    for fn in ("H.txt", "H05.txt", "optimalH.txt", "G05.txt"):
        a, b = oracle.read_pcm(os.path.join(DATA, fn)), ref.read_pcm(os.path.join(DATA, fn))
        assert a.shape == b.shape and (a == b).all()
    H = oracle.read_pcm(os.path.join(DATA, "optimalH.txt"))
    Go, ok1 = oracle.get_orthogonal(H)
    Gr, ok2 = ref.get_orthogonal(H)
    assert ok1 and ok2 and (Go == Gr).all()
    bad = H.copy()
    bad[5] = 0
    assert oracle.get_orthogonal(bad)[1] == ref.get_orthogonal(bad)[1] == False  # noqa: E712
    for seed in (239, 239239239):
        assert (oracle.gen_codewords(Go, seed, 40) == ref.gen_codewords(Gr, seed, 40)).all()
    cw = oracle.gen_codewords(Go, 1, 3)
    for seed in (1, 2, 12345):
        for snr in (-5.0, -0.5, 3.0):
            assert (oracle.transmit(seed, snr, cw[1]) == ref.transmit(seed, snr, cw[1])).all()
    for c in cw:
        assert oracle.is_codeword(H, c) and ref.is_codeword(H, c)
    c = cw[0].copy()
    c[7] ^= 1
    assert not oracle.is_codeword(H, c) and not ref.is_codeword(H, c)


def test_admm_matrix_identical(oracle, ref, matrices):
    for H in matrices.values():
        a, b = oracle.admm_matrix(H), ref.admm_matrix(H)
        for x, y in zip(a, b):
            assert (x == y).all()


@pytest.mark.parametrize("snr", [-3.0, -1.0, 4.0])
def test_decoders_fresh_frames(oracle, ref, matrices, snr):
    # frames NOT in the golden set (seeds 5000+), incl. +4 dB where infinities appear
    H = matrices["H05"]
    G, _ = oracle.get_orthogonal(H)
    cws = oracle.gen_codewords(G, 77, 60)
    y = oracle.transmit_frames(cws, snr, first_seed=5000)
    bo, oo, _ = oracle.bp_decode(H, y, snr, 30, threads=4)
    br, orr, _ = ref.bp_decode(H, y, snr, 30)
    assert (oo == orr).all() and (bo == br).all()
    bo, oo, _ = oracle.qpadmm_decode(H, y, snr, 1.95, 0.5, 60, 1e-5, threads=4)
    br, orr, _ = ref.qpadmm_decode(H, y, snr, 1.95, 0.5, 60, 1e-5)
    assert (oo == orr).all() and (bo == br).all()


def test_experiment_counts(oracle, ref, matrices):
    H = matrices["optimalH"]
    G, _ = oracle.get_orthogonal(H)
    cws = oracle.gen_codewords(G, 239, 120)
    for kind, it, a, m in (("bp", 25, 0, 0), ("qpadmm", 80, 1.2, 0.55)):
        ro = oracle.experiment(kind, H, cws, -2.5, it, a, m)
        rr = ref.experiment(kind, H, cws, -2.5, it, a, m)
        ro.pop("time_sec"), rr.pop("time_sec")
        assert ro == rr
