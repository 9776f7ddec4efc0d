"""CPU: the C-ABI library loads and exports every symbol include/acg_ldpc.h declares, and its
host-side pieces (text format, GetOrtogonal, codeword generator, transmit, QP-ADMM structure)
agree with the oracle.  No compute entry point is called here (no GPU in this container)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "data")


@pytest.fixture(scope="module")
def A():
    import acg_alp_ldpc_amd as A
    A.build()
    return A


def test_exports_every_declared_symbol(A):
    hdr = open(os.path.join(ROOT, "include", "acg_ldpc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(acg_ldpc_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    from acg_alp_ldpc_amd import _lib
    raw = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), "library does not export " + name
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)


def test_abi_struct_sizes(A):
    from acg_alp_ldpc_amd import _lib
    # must match the C layout in include/acg_ldpc.h (x86-64 SysV)
    assert C.sizeof(_lib.Params) == 72   # 2 + 4 doubles + 7 int32 (incl. schedule), padded to 8
    assert C.sizeof(_lib.McCfg) == 56
    assert C.sizeof(_lib.McResult) == 72
    p = _lib.Params()
    A.lib().acg_ldpc_params_default(C.byref(p))
    assert (p.algo, p.max_iter, p.early_exit, p.device) == (0, 50, 1, -1)
    assert (p.alpha, p.mu, p.eps_stop, p.ms_scale) == (1.95, 0.5, 1e-5, 1.0)


def test_text_format_matches_oracle(A, oracle, tmp_path):
    for fn in ("H.txt", "H05.txt", "optimalH.txt", "G05.txt"):
        a = A.read_pcm(os.path.join(DATA, fn)).dense()
        b = oracle.read_pcm(os.path.join(DATA, fn))
        assert a.shape == b.shape and (a == b).all()
    p = tmp_path / "q.txt"
    p.write_text("1,0,1,\n0,21,x1\n  11,10,01  \n")
    H = A.read_pcm(str(p))
    assert H.dense().tolist() == [[1, 0, 1], [0, 1, 1], [1, 0, 1]]
    q = tmp_path / "w.txt"
    H.save_matrix(str(q))
    assert q.read_text() == "1,0,1\n0,1,1\n1,0,1\n"
    with pytest.raises(A.LdpcError):
        A.read_pcm(str(tmp_path / "missing.txt"))
    r = tmp_path / "ragged.txt"
    r.write_text("1,0,1\n1,0\n")
    with pytest.raises(A.LdpcError):
        A.read_pcm(str(r))


def test_generator_codewords_transmit_match_oracle(A, oracle, matrices):
    for name, Hm in matrices.items():
        H = A.ParityCheckMatrix(Hm)
        assert (H.m, H.n, H.E) == (Hm.shape[0], Hm.shape[1], int(Hm.sum()))
        G, ok = H.get_orthogonal()
        Go, oko = oracle.get_orthogonal(Hm)
        assert ok and oko and (G == Go).all()
        cw = A.gen_random_codewords(G, 64, 239239239)
        assert (cw == oracle.gen_codewords(Go, 239239239, 64)).all()
        for c in cw[:8]:
            assert H.is_codeword(c)
        bad = cw[0].copy()
        bad[3] ^= 1
        assert not H.is_codeword(bad)
        for snr in (-3.0, 0.5):
            y = A.transmit_frames(cw, snr)
            assert (y == oracle.transmit_frames(cw, snr)).all()
        y2 = A.transmit_frames(cw, -1.0, first_frame=40, frames=10)   # global frames 40..49, codewords cycle
        for i in range(10):
            assert (y2[i] == oracle.transmit(41 + i, -1.0, cw[(40 + i) % 64])).all()
        assert H.admm_shape() == {k: (int(v) if k in ("n_var", "n_con", "nnz") else float(v))
                                  for k, v in oracle.admm_shape(Hm).items()}
    bad = matrices["H05"].copy()
    bad[7] = 0
    assert A.ParityCheckMatrix(bad).get_orthogonal() == (None, False)
    for snr in (-5, -0.5, 2):
        assert A.llr_variance(snr) == oracle.llr_variance(snr)


def test_no_device_fails_loudly(A):
    if A.device_available():
        pytest.skip("a GPU is present")
    H = A.read_pcm(os.path.join(DATA, "H.txt"))
    with pytest.raises(A.LdpcError, match="no CPU fallback"):
        A.BeliefPropagationDecoder(5).decode(H, np.ones(H.n), 1.0)


def test_shard_ranges_partition():
    from acg_alp_ldpc_amd import shard_range
    for F in (0, 1, 7, 1000, 1 << 20):
        for W in (1, 2, 3, 8):
            spans = [shard_range(F, r, W) for r in range(W)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == F
            for (lo, c), (lo2, _) in zip(spans, spans[1:]):
                assert lo + c == lo2


def test_lds_placement_optimiser(tmp_path):
    """csrc/code.cpp placement_optimise (static placement of QP-ADMM groups / variables against LDS bank conflicts):
    host-only; on a random gather pattern with the v-update's shape it must keep a permutation and cut the modelled
    LDS cycles (sum over lane groups of the busiest bank) by at least a third."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "placement_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "placement_check.cpp"),
                           os.path.join(root, "acg_alp_ldpc_amd", "csrc", "code.cpp"), "-o", exe])
    out = subprocess.run([exe, "544", "300"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    before, after, nsets, _ms = out.stdout.split()
    assert int(after) >= int(nsets)            # one cycle per lane group is the floor
    assert int(after) <= 0.67 * int(before), out.stdout
    # the joint (items + readers, label-preserving) variant used for the BP variable sweep
    out = subprocess.run([exe, "600", "0", "gather"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    before, after, nsets, _ = out.stdout.split()
    assert int(nsets) <= int(after) <= 0.75 * int(before), out.stdout


def test_admm_block_placement_quasi_cyclic(tmp_path):
    """csrc/code.cpp admm_block_placement: for the two quasi-cyclic matrices of the reference (8 x 14 blocks of 20 x 20
    cyclic shifts, optimize_H.cpp:27-63) the constructive tuple placement (mode 2) must be a valid placement, detect
    Z = 20 / tuples of 4, and model no more LDS cycles than the annealed one (mode 1) in every access class; data/H.txt
    is not quasi-cyclic and must fall back to the annealed path unchanged."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "placement_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "placement_check.cpp"),
                           os.path.join(root, "acg_alp_ldpc_amd", "csrc", "code.cpp"), "-o", exe])
    for name in ("H05", "optimalH"):
        out = subprocess.run([exe, "admm", os.path.join(root, "data", name + ".txt"), "256"], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        v = [int(x) for x in out.stdout.split()]
        qc, Z, tup = v[0:3]
        ann, new, ideal = v[3:6], v[6:9], v[9:12]
        wave_ann, wave_new = v[12:16], v[16:20]
        assert (qc, Z, tup) == (1, 20, 4), out.stdout
        assert all(n <= a_ for n, a_ in zip(new, ann)), out.stdout           # never worse than annealing, class by class
        assert new[2] == ideal[2]                                             # V stores: conflict-free by construction
        assert 4 * new[0] + new[1] + new[2] <= 0.8 * (4 * ann[0] + ann[1] + ann[2]), out.stdout   # U rows are read 4x
        assert all(n >= i for n, i in zip(new, ideal))
        assert max(wave_new) < max(wave_ann) and sum(wave_new) == sum(wave_ann)   # v-update trips balanced over the wavefronts
    out = subprocess.run([exe, "admm", os.path.join(root, "data", "H.txt"), "256"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    v = [int(x) for x in out.stdout.split()]
    assert v[0] == 0 and v[3:6] == v[6:9]


def test_valu_mix_tool_prices_the_headline_kernels():
    """tools/valu_mix.py (used by bench.py for roofline.frac): the static instruction mix of the probed kernels is found in the
    built library and lies between the all-full-rate and all-half-rate prices"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import valu_mix
    if not os.path.exists(valu_mix.LLVM + "/llvm-objdump"):
        pytest.skip("no llvm-objdump in this image")
    mix = valu_mix.static_mix({"bp": "bp_fused_kernel<float, 8, 32, 0, false, true, 12, false>", "admm": "admm_block_kernel<double, false, 3, true>"})
    assert set(mix) == {"bp", "admm"}
    for m in mix.values():
        assert m["valu_static"] > 500 and 2.0 <= m["cycles_per_non_transcendental"] <= 4.0
    assert mix["bp"]["transcendental"] > 50 and mix["admm"]["transcendental"] == 0
    assert mix["admm"]["half_rate"] > mix["admm"]["full_rate"]   # fp64 arithmetic is all half-rate
