"""the C++ drivers that replace the reference's main.cpp / qpadmm_params.cpp /
optimize_H.cpp loops.  CPU: they build with plain g++ and the host-only paths work.  GPU: results against the
oracle / the reference's known answers."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "drivers", "bin")
DATA = os.path.join(ROOT, "data")


@pytest.fixture(scope="module")
def drivers():
    import acg_alp_ldpc_amd as A
    A.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools", "drivers")], stdout=subprocess.DEVNULL)
    return BIN


def test_drivers_build_and_qc_roundtrip(drivers, oracle):
    for exe in ("acg_eval", "acg_qpadmm_params", "acg_optimize_h"):
        assert os.access(os.path.join(drivers, exe), os.X_OK)
    for fn in ("H05.txt", "optimalH.txt"):
        out = subprocess.run([os.path.join(drivers, "acg_optimize_h"), "--init", os.path.join(DATA, fn), "--check-qc"],
                             capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stderr
        lines = out.stdout.strip().splitlines()
        assert lines[0] == "Z=20 R=8 C=14"           # 8 x 14 protograph of 20 x 20 circulants
        tab = np.array([[int(x) for x in l.split()] for l in lines[1:]])
        H = oracle.read_pcm(os.path.join(DATA, fn))
        rebuilt = np.zeros_like(H)
        for i in range(8):
            for j in range(14):
                if tab[i, j] >= 0:
                    for k in range(20):
                        rebuilt[i * 20 + k, j * 20 + (tab[i, j] + k) % 20] = 1
        assert (rebuilt == H).all()
    # a non-QC matrix is refused
    out = subprocess.run([os.path.join(drivers, "acg_optimize_h"), "--init", os.path.join(DATA, "H.txt"), "--Z", "16",
                          "--check-qc"], capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 or "Z=16" in out.stdout


def test_optimize_driver_proposal_sequence_is_the_references(drivers):
    """N3 pin (CPU): the first 64 proposals of acg_optimize_h — block row, block column, presence, shift — equal those of the
    reference's PermutationsMatrix::random_permute (optimize_H.cpp:66-75) under std::mt19937(239) (optimize_H.cpp:132), for the
    all-rejected and the all-accepted chain, on H05 and optimalH.  The fixture was written by oracle/make_golden_optimize.py
    from the real reference compiled as it lies (oracle/ref_optimize_shim.cpp).  Acceptance itself cannot be pinned (it rides
    on a 200-thread seed race)."""
    import json
    fix = json.load(open(os.path.join(ROOT, "tests", "golden", "optimize_h_proposals.json")))
    assert len(fix["cases"]) == 8
    for case in fix["cases"]:
        cmd = [os.path.join(drivers, "acg_optimize_h"), "--init", os.path.join(DATA, case["matrix"]), "--Z", str(case["Z"]),
               "--seed", str(case["seed"]), "--dump-proposals", str(len(case["proposals"]))]
        if case["accept_all"]:
            cmd.append("--accept-all")
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stderr
        got = [[int(x) for x in line.split()] for line in out.stdout.strip().split("\n")]
        assert got == case["proposals"], (case["matrix"], case["seed"], case["accept_all"])
    # the two chains really differ (a proposal on a block mutated earlier sees the mutated presence)
    assert fix["cases"][0]["proposals"] != fix["cases"][1]["proposals"]


def test_reference_shim_reproduces_the_fixture():
    """the fixture against the real reference where it is available (this container); skipped on the GPU box"""
    import ctypes as C
    import json
    import numpy as np
    so = os.path.join(ROOT, "oracle", "_ref", "libacg_ref_opt.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libacg_ref_opt.so not built (no /root/reference here)")
    lib = C.CDLL(so)
    lib.ref_opt_proposals.argtypes = [C.c_char_p, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_void_p]
    fix = json.load(open(os.path.join(ROOT, "tests", "golden", "optimize_h_proposals.json")))
    for case in fix["cases"]:
        out = np.zeros((len(case["proposals"]), 4), dtype=np.int32)
        rc = lib.ref_opt_proposals(os.path.join(DATA, case["matrix"]).encode(), case["Z"], case["seed"], len(out), case["accept_all"],
                                   out.ctypes.data_as(C.c_void_p))
        assert rc == 0 and out.tolist() == case["proposals"]


@pytest.mark.gpu
def test_eval_driver_reproduces_reference_known_answers(drivers, tmp_path):
    """main.cpp loop with the reference's exact frames: H05, BP(50) and QP-ADMM(1.95,0.5,100) at -2 dB, 1000 frames,
    codewords GetOrtogonal + mt19937(239239239) -> FER 0.085 / 0.340, AvgHamming 36.302 (known_answers.json)."""
    out = str(tmp_path / "report.csv")
    r = subprocess.run([os.path.join(drivers, "acg_eval"), "--H", os.path.join(DATA, "H05.txt"), "--snrs", "-2,0",
                        "--tests", "1000", "--bp-iters", "50", "--alpha", "1.95", "--mu", "0.5", "--admm-iters", "100",
                        "--noise", "host", "--out", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    rows = open(out).read().strip().splitlines()
    assert rows[0] == "Method,SNR,Sigma,FER,Time,AvgHamming,AvgHammingCorrect,AvgHammingWrong"   # main.cpp:48
    tab = {(x.split(",")[0], float(x.split(",")[1])): [float(v) for v in x.split(",")[2:]] for x in rows[1:]}
    assert abs(tab[("BP", -2.0)][0] - 0.890194695688) < 1e-12          # Sigma pin, reports/report_H05.csv:8
    assert tab[("BP", -2.0)][1] == pytest.approx(0.085, abs=1e-12)
    assert tab[("BP", 0.0)][1] == 0.0
    assert tab[("QP-ADMM", -2.0)][1] == pytest.approx(0.340, abs=1e-12)
    assert tab[("QP-ADMM", 0.0)][1] == pytest.approx(0.003, abs=1e-12)
    assert tab[("BP", -2.0)][3] == pytest.approx(36.302, abs=1e-9)      # AvgHamming is decoder independent
    assert tab[("QP-ADMM", -2.0)][3] == pytest.approx(36.302, abs=1e-9)
    assert "Algo: BP" in r.stdout and "Algo: QP-ADMM" in r.stdout


@pytest.mark.gpu
def test_eval_driver_minsum_rows(drivers, oracle, tmp_path):
    """the optional min-sum rows of acg_eval (build-added variant, parity unpinned): with the reference's host-noise frames the FER
    of the layered decoder equals what the repo's numpy restatement gets on the same 1000 frames, and the flooding row's FER is
    the oracle min-sum's"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from layered_ref import layered_minsum
    import acg_alp_ldpc_amd as A
    Hm = oracle.read_pcm(os.path.join(DATA, "H05.txt"))
    G, _ = oracle.get_orthogonal(Hm)
    cws = oracle.gen_codewords(G, 239239239, 1000)
    y = oracle.transmit_frames(cws, -2.0, first_seed=1)
    _, _, layers = A.ParityCheckMatrix(Hm).layers()
    rb, rok, _ = layered_minsum(Hm, layers, y, -2.0, 25, 0.75)
    fer_lay = 1 - ((rok == 1) & (rb == cws).all(axis=1)).mean()
    ob, ook, _ = oracle.minsum_decode(Hm, y, -2.0, 50, 0.75)
    fer_flo = 1 - ((ook == 1) & (ob == cws).all(axis=1)).mean()
    for extra, want in ((["--minsum-iters", "25", "--layered"], fer_lay), (["--minsum-iters", "50"], fer_flo)):
        csv = str(tmp_path / "r.csv")
        r = subprocess.run([os.path.join(drivers, "acg_eval"), "--H", os.path.join(DATA, "H05.txt"), "--snrs", "-2", "--tests", "1000", "--noise", "host",
                            "--no-bp", "--no-admm", "--out", csv] + extra, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        row = open(csv).read().strip().split("\n")[1].split(",")
        assert row[0] == "MS" and abs(float(row[3]) - want) < 1e-9, (extra, row, want)


@pytest.mark.gpu
def test_grid_search_driver_matches_oracle(drivers, oracle):
    """qpadmm_params.cpp loop on a 3 x 3 sub-grid, 200 frames: every FER and the winner equal the oracle's"""
    H = oracle.read_pcm(os.path.join(DATA, "optimalH.txt"))
    G, _ = oracle.get_orthogonal(H)
    cws = oracle.gen_codewords(G, 239, 200)
    r = subprocess.run([os.path.join(drivers, "acg_qpadmm_params"), "--H", os.path.join(DATA, "optimalH.txt"),
                        "--tests", "200", "--iters", "300", "--alpha", "0.8,1.6,3", "--mu", "0.3,0.7,3", "--noise", "host"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    got = {}
    for l in r.stderr.splitlines():
        if l.startswith("alpha="):
            a, m, f = l.replace("alpha=", "").replace(" mu=", "").replace(": fer=", ",").split(",")
            got[(round(float(a), 5), round(float(m), 5))] = float(f)
    assert len(got) == 9
    best = (2.0, None)
    for a in (0.8, 1.2, 1.6):
        for m in (0.3, 0.5, 0.7):
            if 4.0 * m <= a:
                fer = 1.0
            else:
                res = oracle.experiment("qpadmm", H, cws, -3.0, 300, a, m, 1e-5)
                fer = (res["total"] - res["correct"]) / res["total"]
            assert got[(a, m)] == pytest.approx(fer, abs=1e-5), (a, m)
            if fer < best[0]:
                best = (fer, (a, m))
    assert ("alpha=%.5f" % best[1][0]) in r.stdout and ("mu=%.5f" % best[1][1]) in r.stdout


@pytest.mark.gpu
def test_optimize_h_driver_runs_and_only_accepts_improvements(drivers, oracle, tmp_path):
    out = str(tmp_path / "opt.txt")
    r = subprocess.run([os.path.join(drivers, "acg_optimize_h"), "--init", os.path.join(DATA, "H05.txt"), "--iters", "6",
                        "--tests", "300", "--admm-iters", "200", "--noise", "host", "--out", out],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    H = oracle.read_pcm(os.path.join(DATA, "H05.txt"))
    G, _ = oracle.get_orthogonal(H)
    cws = oracle.gen_codewords(G, 239, 300)
    res = oracle.experiment("qpadmm", H, cws, -3.0, 200, 1.95, 0.5, 1e-5)
    fer0 = (res["total"] - res["correct"]) / res["total"]
    assert lines[0] == "initial FER=%.5f" % fer0          # FER(H) of optimize_H.cpp:16-25, same frames as the oracle
    assert sum(l.startswith("\tproposal") for l in lines) == 6
    cur = fer0
    for l in lines[1:]:
        if l.startswith("accept"):
            f = float(l.split("=")[1])
            assert f < cur
            cur = f
    if os.path.exists(out):                                # an accepted proposal was saved in the reference text format
        Hn = oracle.read_pcm(out)
        assert Hn.shape == H.shape
        Gn, ok = oracle.get_orthogonal(Hn)
        assert ok
        cw2 = oracle.gen_codewords(Gn, 239, 300)
        res = oracle.experiment("qpadmm", Hn, cw2, -3.0, 200, 1.95, 0.5, 1e-5)
        assert (res["total"] - res["correct"]) / res["total"] == pytest.approx(cur, abs=1e-5)


@pytest.mark.gpu
def test_eval_driver_multi_handle_sharding(drivers, tmp_path):
    """--gpus 3 (three handles + three host threads, folded onto the one device of the test box): contiguous global
    frame ranges + merged counters give the same report as one handle, for host and for device noise"""
    rows = {}
    for tag, extra in (("one", []), ("three", ["--gpus", "3", "--device-count", "1"])):
        for noise in ("host", "device"):
            out = str(tmp_path / ("r_%s_%s.csv" % (tag, noise)))
            r = subprocess.run([os.path.join(drivers, "acg_eval"), "--H", os.path.join(DATA, "H05.txt"), "--snrs", "-2",
                                "--tests", "3001", "--bp-iters", "50", "--alpha", "1.95", "--mu", "0.5", "--admm-iters",
                                "100", "--noise", noise, "--out", out] + extra, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr
            rows[(tag, noise)] = [[x.split(",")[i] for i in (0, 1, 2, 3, 5, 6, 7)] for x in open(out).read().strip().splitlines()[1:]]
    for noise in ("host", "device"):
        assert rows[("one", noise)] == rows[("three", noise)]   # every column except Time
