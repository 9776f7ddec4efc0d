"""Layered min-sum schedule (SURVEY §8(f) N4 on the BP side; csrc/bp_layered.hip).

A different algorithm from the reference's flooding sum-product (algo/bp.h:183-199): PARITY IS FER-LEVEL ONLY, and min-sum
itself is build-added (parity unpinned).  What is checked:
  CPU  the layering: every check in exactly one layer, no variable twice in a layer, one degree per layer; H05 / optimalH
       (8 x 14 arrays of 20 x 20 circulants, optimize_H.cpp:27-63) -> their 8 block rows, 20 lanes per frame
  GPU  the kernel against the repo's own numpy restatement, word for word (tests/layered_ref.py)
  GPU  FER at 25 layered iterations <= FER at 50 flooding iterations (+ binomial slack) over >= 10^6 device-noise frames
       on H05 and optimalH at -2 / -1 dB; every word the layered decoder returns with ok = 1 is a codeword
  GPU  fixed-work mode latches the same outputs as early exit; the flooding kernels are untouched (the rest of the suite)"""
import os

import numpy as np
import pytest

from layered_ref import layered_minsum, layered_sumproduct

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")


def _check_layers(Hm, G, layers):
    seen = np.zeros(Hm.shape[0], dtype=int)
    for layer in layers:
        ids = layer[layer >= 0]
        assert len(ids) > 0 and len(ids) <= G
        seen[ids] += 1
        sub = Hm[ids]
        assert (sub.sum(axis=0) <= 1).all()              # no variable twice in a layer
        assert len(set(sub.sum(axis=1))) == 1            # one degree per layer
        assert (layer[len(ids):] == -1).all()            # occupied lanes first
    deg = Hm.sum(axis=1)
    assert (seen[deg > 0] == 1).all() and (seen[deg == 0] == 0).all()


def test_layers_of_the_reference_matrices_are_their_block_rows(matrices):
    import acg_alp_ldpc_amd as A
    for name in ("H05", "optimalH"):
        G, Z, layers = A.ParityCheckMatrix(matrices[name]).layers()
        assert (G, Z, layers.shape) == (20, 20, (8, 20))
        assert (layers == np.arange(160).reshape(8, 20)).all()
        _check_layers(matrices[name], G, layers)


def test_layers_of_other_matrices_are_conflict_free(matrices):
    import acg_alp_ldpc_amd as A
    cases = [matrices["H"], A.regular_ldpc(48, 96, 3, 6, seed=3), A.regular_ldpc(300, 600, 3, 6, seed=4)]
    ragged = A.regular_ldpc(40, 80, 3, 6, seed=5).copy()
    ragged[0, :] = 0          # an empty check
    ragged[1, np.nonzero(ragged[1])[0][:3]] = 0   # a degree-3 check among degree-6 ones
    cases.append(ragged)
    for Hm in cases:
        G, Z, layers = A.ParityCheckMatrix(Hm).layers()
        assert G in (16, 20, 32, 64)
        _check_layers(np.asarray(Hm), G, layers)


def test_layered_restatement_decodes(oracle, matrices):
    """the numpy restatement itself: clean codewords come back in one iteration, noisy ones at +1 dB mostly decode"""
    Hm = matrices["H05"]
    G, _ = oracle.get_orthogonal(Hm)
    cws = oracle.gen_codewords(G, 11, 64)
    layers = np.arange(160).reshape(8, 20)
    y = 1.0 - 2.0 * cws.astype(np.float64)
    bits, ok, iters = layered_minsum(Hm, layers, y, 1.0, 10, 0.75)
    assert ok.all() and (bits == cws).all() and (iters == 1).all()
    y = oracle.transmit_frames(cws, 1.0, first_seed=1)
    bits, ok, iters = layered_minsum(Hm, layers, y, 1.0, 25, 0.75)
    assert ok.mean() > 0.9 and (bits[ok == 1] == cws[ok == 1]).mean() > 0.99
    assert all(oracle.is_codeword(Hm, b) for b in bits[ok == 1])


def test_layered_sumproduct_restatement_against_the_oracle(oracle, matrices):
    """CPU: the float64 restatement of the layered sum-product — clean codewords stop after one quiet iteration; on 200 noisy
    frames at -2 dB its 25 iterations decode at least as many frames as the oracle's (= the reference's) 50 flooding iterations,
    minus two frames of slack, and every word it returns satisfies H"""
    Hm = matrices["H05"]
    G, _ = oracle.get_orthogonal(Hm)
    cws = oracle.gen_codewords(G, 12, 200)
    layers = np.arange(160).reshape(8, 20)
    y = 1.0 - 2.0 * cws.astype(np.float64)
    bits, ok, iters = layered_sumproduct(Hm, layers, y, 0.0, 10)
    assert ok.all() and (bits == cws).all() and (iters == 1).all()
    y = oracle.transmit_frames(cws, -2.0, first_seed=500)
    bits, ok, iters = layered_sumproduct(Hm, layers, y, -2.0, 25)
    ob, ook, oit = oracle.bp_decode(Hm, y, -2.0, 50)
    good = (ok == 1) & (bits == cws).all(axis=1)
    ogood = (ook == 1) & (ob == cws).all(axis=1)
    assert good.sum() >= ogood.sum() - 2, (int(good.sum()), int(ogood.sum()))
    assert all(oracle.is_codeword(Hm, b) for b in bits[ok == 1])
    assert iters[ok == 1].mean() < 0.8 * oit[ook == 1].mean()


# ------------------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def A():
    import acg_alp_ldpc_amd as A
    return A


@pytest.mark.gpu
@pytest.mark.parametrize("msg", ["f32", "f16"])
@pytest.mark.parametrize("name,snr", [("H05", -2.0), ("H05", 0.5), ("optimalH", -2.0), ("H", 1.0)])
def test_layered_kernel_equals_restatement(A, oracle, matrices, name, snr, msg):
    """words, flags and iteration counts identical to the numpy restatement (the repo's own: parity unpinned), with the
    reference's stopping semantics and in fixed-work mode, for 25, 3 and 0 iterations; messages stored in fp32 or (precision
    PREC_F16) in half precision"""
    Hm = matrices[name]
    H = A.ParityCheckMatrix(Hm)
    G, _ = oracle.get_orthogonal(Hm)
    cws = oracle.gen_codewords(G, 5, 600)
    y = oracle.transmit_frames(cws, snr, first_seed=1)
    _, _, layers = H.layers()
    for it in (25, 3, 0):
        rb, rok, rit = layered_minsum(Hm, layers, y, snr, it, 0.75, np.float16 if msg == "f16" else np.float32)
        for ee in (True, False):
            dec = A.MinSumDecoder(it, 0.75, schedule=A.SCHEDULE_LAYERED, early_exit=ee,
                                  precision=A.PREC_F16 if msg == "f16" else A.PREC_DEFAULT)
            bits, ok, iters = dec.decode_batch(H, y, snr)
            assert "bp_layered_kernel" in dec.describe(H) and "layered" in dec.describe(H)
            dec.close()
            assert (ok == rok).all(), (name, snr, it, ee, int((ok != rok).sum()))
            assert (bits == rb).all() and (iters == rit).all(), (name, snr, it, ee)
    assert 0.02 < 1 - rok.mean() or snr > 0     # (the 25-iteration run at -2 dB does fail some frames: both paths covered)


@pytest.mark.gpu
def test_layered_ragged_batches_and_float_symbols(A, oracle, matrices):
    Hm = matrices["optimalH"]
    H = A.ParityCheckMatrix(Hm)
    G, _ = oracle.get_orthogonal(Hm)
    cws = oracle.gen_codewords(G, 9, 257)
    y = oracle.transmit_frames(cws, -1.0, first_seed=50)
    _, _, layers = H.layers()
    dec = A.MinSumDecoder(20, 0.8, schedule=A.SCHEDULE_LAYERED)
    for F in (1, 2, 3, 59, 60, 61, 257):
        rb, rok, rit = layered_minsum(Hm, layers, y[:F], -1.0, 20, 0.8)
        bits, ok, iters = dec.decode_batch(H, y[:F], -1.0)
        assert (ok == rok).all() and (bits == rb).all() and (iters == rit).all(), F
    # float32 symbols take the (double) y * (2 / sigma^2) path: same decisions as the restatement fed with the rounded symbols
    y32 = y.astype(np.float32)
    bits, ok, iters = dec.decode_batch(H, y32, -1.0)
    rb, rok, rit = layered_minsum(Hm, layers, y32.astype(np.float64), -1.0, 20, 0.8)
    assert (ok == rok).mean() > 0.995 and (bits[ok == rok] == rb[ok == rok]).all(axis=1).mean() > 0.995
    dec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("algo", ["minsum", "bp"])
@pytest.mark.parametrize("prec", ["f32", "f16"])
def test_layered_monte_carlo_fused_equals_unfused(A, matrices, algo, prec, monkeypatch):
    """the Monte-Carlo instance of the layered kernel (noise generated and words classified in the kernel) gives the same seven
    counters as AWGN kernel -> decode kernel -> classification kernel on the same global frames, for any split into shards"""
    H = A.ParityCheckMatrix(matrices["H05"])
    G, _ = H.get_orthogonal()
    cws = A.gen_random_codewords(G, 1000, 239239239)
    pr = A.PREC_F16 if prec == "f16" else A.PREC_DEFAULT

    def make():
        if algo == "bp":
            return A.BeliefPropagationDecoder(20, schedule=A.SCHEDULE_LAYERED, precision=pr)
        return A.MinSumDecoder(20, 0.75, schedule=A.SCHEDULE_LAYERED, precision=pr)
    F = 50000
    for snr in (-2.0, 1.0):
        d = make()
        fused = A.run_experiment(d, cws, H, snr, frames=F, noise="device", seed=9).as_vector()
        parts = sum(A.run_experiment(d, cws, H, snr, frames=c, first_frame=lo, noise="device", seed=9).as_vector()
                    for lo, c in ((0, 12345), (12345, 1), (12346, F - 12346)))
        d.close()
        monkeypatch.setenv("ACG_LAY_UNFUSED_MC", "1")
        d = make()
        unfused = A.run_experiment(d, cws, H, snr, frames=F, noise="device", seed=9).as_vector()
        d.close()
        monkeypatch.delenv("ACG_LAY_UNFUSED_MC")
        assert (fused == unfused).all(), (algo, prec, snr, fused, unfused)
        assert (fused == parts).all(), (algo, prec, snr, fused, parts)
        assert fused[2] == F and fused[3] == fused[4] + fused[5] and fused[0] > 0


@pytest.mark.gpu
def test_layered_ragged_graph(A, oracle):
    """a matrix that is neither quasi-cyclic nor regular: an empty check, a degree-3 check among degree-6 ones, isolated
    variables, a degree-1 check — greedy-coloured layers of several degrees with partly filled lanes"""
    Hm = A.regular_ldpc(40, 80, 3, 6, seed=5).copy()
    Hm[0, :] = 0
    Hm[1, np.nonzero(Hm[1])[0][:3]] = 0
    Hm[2, np.nonzero(Hm[2])[0][1:]] = 0          # degree 1
    Hm[:, 7] = 0                                  # an isolated variable
    H = A.ParityCheckMatrix(Hm)
    G, Z, layers = H.layers()
    assert Z == 0 and len(set(Hm[l[l >= 0]].sum(axis=1)[0] for l in layers)) >= 3
    y = oracle.transmit_frames(np.zeros((400, 80), dtype=np.uint8), 1.0, first_seed=9)
    for prec, dt in ((A.PREC_DEFAULT, np.float32), (A.PREC_F16, np.float16)):
        rb, rok, rit = layered_minsum(Hm, layers, y, 1.0, 15, 0.8, dt)
        dec = A.MinSumDecoder(15, 0.8, schedule=A.SCHEDULE_LAYERED, precision=prec)
        bits, ok, iters = dec.decode_batch(H, y, 1.0)
        dec.close()
        assert (ok == rok).all() and (bits == rb).all() and (iters == rit).all()
        assert 0 < ok.sum() < len(ok) or ok.all()


@pytest.mark.gpu
def test_layered_refuses_what_it_is_not(A, matrices):
    H = A.ParityCheckMatrix(matrices["H05"])
    y = np.ones((1, 280))
    with pytest.raises(A.LdpcError):
        A.MinSumDecoder(10, 0.75, schedule=A.SCHEDULE_LAYERED, engine=A.ENGINE_STREAMED).decode_batch(H, y, 0.0)
    with pytest.raises(A.LdpcError):
        A.MinSumDecoder(10, 0.75, schedule=A.SCHEDULE_LAYERED, precision=A.PREC_F64).decode_batch(H, y, 0.0)
    # the arithmetic-addressing instance (developer A/B switch) gives the same words as the table-driven one
    import os
    yy = np.random.default_rng(1).normal(1.0, 0.9, size=(300, 280))
    ref = A.MinSumDecoder(12, 0.75, schedule=A.SCHEDULE_LAYERED).decode_batch(H, yy, -1.0)
    os.environ["ACG_LAY_ARITH"] = "1"
    try:
        got = A.MinSumDecoder(12, 0.75, schedule=A.SCHEDULE_LAYERED).decode_batch(H, yy, -1.0)
    finally:
        del os.environ["ACG_LAY_ARITH"]
    assert all((a == b).all() for a, b in zip(ref, got))


@pytest.mark.gpu
@pytest.mark.parametrize("name,snr", [("H05", -2.0), ("optimalH", -1.0), ("H", 1.5)])
def test_layered_sumproduct_agrees_with_float64_restatement(A, oracle, matrices, name, snr):
    """the sum-product variant of the layered kernel (the reference's check rule bp.h:49-57 in the layered order; fp32, the
    flooding kernels' phi) against a float64 numpy restatement with the exact phi: an agreement RATE — >= 99 % of the frames with
    the same flag, word and iteration count (knife edges between fp32 and fp64 move single frames), the rest same FER"""
    Hm = matrices[name]
    H = A.ParityCheckMatrix(Hm)
    G, _ = oracle.get_orthogonal(Hm)
    cws = oracle.gen_codewords(G, 6, 500)
    y = oracle.transmit_frames(cws, snr, first_seed=77)
    _, _, layers = H.layers()
    rb, rok, rit = layered_sumproduct(Hm, layers, y, snr, 25)
    for prec in (A.PREC_DEFAULT, A.PREC_F16):
        dec = A.BeliefPropagationDecoder(25, schedule=A.SCHEDULE_LAYERED, precision=prec)
        bits, ok, iters = dec.decode_batch(H, y, snr)
        assert "bp_layered_kernel" in dec.describe(H) and dec.describe(H).startswith("sum-product")
        dec.close()
        same = (ok == rok) & (bits == rb).all(axis=1)
        bar = 0.99 if prec == A.PREC_DEFAULT else 0.97
        assert same.mean() >= bar, (name, prec, same.mean())
        assert (iters[same] == rit[same]).mean() >= bar
        assert abs(ok.mean() - rok.mean()) <= 0.01
        good = ok == 1
        assert all(oracle.is_codeword(Hm, b) for b in bits[good][:100])      # every ok = 1 word satisfies H


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["H05", "optimalH"])
def test_layered_sumproduct_25_matches_the_references_fer(A, matrices, name):
    """2^20 device-noise frames per point: FER(layered sum-product, 25 iterations) <= FER(the reference's flooding sum-product,
    50 iterations) + binomial slack at -2 and -1 dB — the reference's frame error rate at about half its iterations.  FER-level
    parity is all a layered schedule can have."""
    H = A.ParityCheckMatrix(matrices[name])
    G, okG = H.get_orthogonal()
    cws = A.gen_random_codewords(G, 4096, 239239239)
    F = 1 << 20
    for snr in (-2.0, -1.0):
        lay = A.BeliefPropagationDecoder(25, schedule=A.SCHEDULE_LAYERED)
        spa = A.BeliefPropagationDecoder(50)
        rl = A.run_experiment(lay, cws, H, snr, frames=F, noise="device", seed=3)
        rs = A.run_experiment(spa, cws, H, snr, frames=F, noise="device", seed=3)
        lay.close()
        spa.close()
        assert rl.total == rs.total == F and rl.sum_hamming == rs.sum_hamming      # the same frames
        fl, fs = rl.FER(), rs.FER()
        slack = 4.0 * np.sqrt(max(fs * (1 - fs), 1e-6) / F) * np.sqrt(2)
        print("%s %+.1f dB: FER layered sum-product-25 %.5f  flooding sum-product-50 (the reference's algorithm) %.5f; mean iterations %.2f / %.2f; pseudo %d / %d"
              % (name, snr, fl, fs, rl.mean_iters(), rs.mean_iters(), rl.pseudo, rs.pseudo))
        assert fl <= fs + slack, (name, snr, fl, fs)
        assert rl.mean_iters() < 0.85 * rs.mean_iters()


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "f16"])
@pytest.mark.parametrize("name", ["H05", "optimalH"])
def test_layered_25_not_worse_than_flooding_50(A, matrices, name, prec):
    """>= 2^20 device-noise frames per point: FER(layered min-sum, 25 iterations) <= FER(flooding min-sum, 50) + binomial
    slack, at -2 and -1 dB; every ok = 1 word is a codeword (pseudo-codewords are counted by the classification kernel,
    which recomputes the syndrome); the sum-product FER of the reference's own algorithm is printed beside them"""
    H = A.ParityCheckMatrix(matrices[name])
    G, okG = H.get_orthogonal()
    cws = A.gen_random_codewords(G, 4096, 239239239)
    F = 1 << 20
    for snr in (-2.0, -1.0):
        lay = A.MinSumDecoder(25, 0.75, schedule=A.SCHEDULE_LAYERED, precision=A.PREC_F16 if prec == "f16" else A.PREC_DEFAULT)
        flo = A.MinSumDecoder(50, 0.75)
        spa = A.BeliefPropagationDecoder(50)
        rl = A.run_experiment(lay, cws, H, snr, frames=F, noise="device", seed=3)
        rf = A.run_experiment(flo, cws, H, snr, frames=F, noise="device", seed=3)
        rs = A.run_experiment(spa, cws, H, snr, frames=F, noise="device", seed=3)
        for d in (lay, flo, spa):
            d.close()
        assert rl.total == rf.total == F and rl.sum_hamming == rf.sum_hamming      # the same frames
        fl, ff = rl.FER(), rf.FER()
        slack = 4.0 * np.sqrt(max(ff * (1 - ff), 1e-6) / F) * np.sqrt(2)
        print("%s %s %+.1f dB: FER layered-25 %.5f  flooding-50 %.5f  sum-product-50 %.5f; mean iterations %.2f / %.2f / %.2f; pseudo %d / %d"
              % (name, prec, snr, fl, ff, rs.FER(), rl.mean_iters(), rf.mean_iters(), rs.mean_iters(), rl.pseudo, rf.pseudo))
        assert fl <= ff + slack, (name, snr, fl, ff)
        # fewer sweeps: about half while a frame is still moving, plus the one quiet round that proves convergence
        assert rl.mean_iters() < 0.85 * rf.mean_iters()
