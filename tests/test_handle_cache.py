"""The analysed-graph caches of the host mirrors are bounded (VERDICT r02 #8 / ADVICE): the reference's optimize_H loop hands
a NEW H per proposal to ONE shared decoder (optimize_H.cpp:16-25,89-104, experiment.h:101), so an unbounded cache would
collect one device handle (work ring, 64 events, slabs) per proposal.  Both mirrors keep the 8 most recently used."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _threads():
    for line in open("/proc/self/status"):
        if line.startswith("Threads:"):
            return int(line.split()[1])
    return -1


def test_python_key_is_content_not_hash():
    """CPU: the dense-array key carries the packed bits themselves (equality, not hash() alone)"""
    from acg_alp_ldpc_amd.decoder import BeliefPropagationDecoder
    d = BeliefPropagationDecoder(5)
    a = np.zeros((4, 8), dtype=np.uint8)
    b = a.copy()
    b[3, 7] = 1
    ka, kb = d._key(a), d._key(b)
    assert ka != kb and ka == d._key(a.astype(np.int64)) and ka == d._key(np.ascontiguousarray(a))
    assert isinstance(ka[2], bytes) and len(ka[2]) == 4      # 32 bits, packed
    assert d._key(a.reshape(8, 4)) != ka                       # same bits, other shape


@pytest.mark.gpu
@pytest.mark.parametrize("algo", ["bp", "qpadmm"])
def test_one_decoder_64_matrices_bounded_handles(oracle, algo):
    """ONE decoder object, 64 distinct H: at most MAX_HANDLES live device handles, a bounded thread count, and the
    results for the first and the last H — and for the first H again after it was evicted — equal to the oracle"""
    import acg_alp_ldpc_amd as A
    assert A.device_available()
    dec = A.BeliefPropagationDecoder(20) if algo == "bp" else A.QPADMMDecoder(1.2, 0.55, 60, 1e-5, fast_setup=True)
    cap = dec.max_handles
    assert cap == 8
    snr = 2.0
    mats, ys = [], []
    for s in range(64):
        Hm = A.regular_ldpc(48, 96, 3, 6, seed=100 + s)
        mats.append(Hm)
        ys.append(oracle.transmit_frames(np.zeros((4, 96), dtype=np.uint8), snr, first_seed=1 + 4 * s))
    t0 = None
    out = []
    for s, (Hm, y) in enumerate(zip(mats, ys)):
        out.append(dec.decode_batch(Hm, y, snr))
        assert dec.live_handles() <= cap
        if s == 8:
            t0 = _threads()
    assert dec.live_handles() == cap
    assert _threads() <= t0 + 2, (t0, _threads())       # no threads collected per handle
    again = dec.decode_batch(mats[0], ys[0], snr)         # evicted long ago: re-analysed, same answer
    for s, got in ((0, out[0]), (63, out[63]), (0, again)):
        if algo == "bp":
            ob, ook, oit = oracle.bp_decode(mats[s], ys[s], snr, 20)
        else:
            ob, ook, oit = oracle.qpadmm_decode(mats[s], ys[s], snr, 1.2, 0.55, 60, 1e-5)
        assert (got[1] == ook).all() and (got[0] == ob).all() and (got[2] == oit).all(), (algo, s)
    # a large batch through the same object starts the process-wide host pool ONCE, whatever the number of handles
    big = oracle.transmit_frames(np.zeros((1, 96), dtype=np.uint8), snr, first_seed=7).repeat(8192, axis=0)
    dec.decode_batch(mats[1], big, snr)
    t1 = _threads()
    dec.decode_batch(mats[2], big, snr)
    dec.decode_batch(mats[3], big, snr)
    assert _threads() == t1
    dec.close()
    assert dec.live_handles() == 0


@pytest.mark.gpu
def test_one_decoder_many_matrices_from_threads():
    """4 host threads, 12 distinct H, ONE decoder object with room for 8 handles (ctypes calls release the GIL): a handle in use
    by one thread is pinned, never destroyed under it; every result equals the single-threaded one"""
    import threading
    import acg_alp_ldpc_amd as A
    rng = np.random.default_rng(3)
    mats = [A.regular_ldpc(48, 96, 3, 6, seed=500 + s) for s in range(12)]
    ys = [rng.normal(1.0, 0.7, size=(64, 96)) for _ in mats]
    ref = A.BeliefPropagationDecoder(15)
    want = [ref.decode_batch(Hm, y, 1.0) for Hm, y in zip(mats, ys)]
    ref.close()
    dec = A.BeliefPropagationDecoder(15)
    errs = []

    def work(t):
        try:
            for rep in range(6):
                for s in range(t, 12, 4) if rep % 2 == 0 else range(12):
                    got = dec.decode_batch(mats[s], ys[s], 1.0)
                    assert all((a == b).all() for a, b in zip(got, want[s])), (t, rep, s)
                    assert dec.live_handles() <= dec.max_handles + 4
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs[0]
    assert dec.live_handles() <= dec.max_handles
    dec.close()


@pytest.mark.gpu
def test_eviction_waits_for_asynchronous_launches():
    """acg_ldpc_decode_batch_dev is asynchronous on the caller's stream: launching through ONE decoder object on 12 distinct H
    back to back evicts (destroys) the first handles while their kernels may still be queued — destroy waits for the launches of
    the handle (the stop events of its work-ring slots) before it frees the tables.  Results equal the synchronous ones."""
    import torch
    import acg_alp_ldpc_amd as A
    rng = np.random.default_rng(11)
    F, n = 32768, 96
    mats = [A.ParityCheckMatrix(A.regular_ldpc(48, 96, 3, 6, seed=900 + s)) for s in range(12)]
    ys = [rng.normal(1.0, 0.8, size=(F, n)).astype(np.float32) for _ in mats]
    ref = A.BeliefPropagationDecoder(40, early_exit=False)
    want = []
    for H, y in zip(mats, ys):
        b, ok, it = ref.decode_batch(H, y, 1.0)
        want.append((np.packbits(b, axis=1, bitorder="little"), ok, it))
        ref.close()
    dec = A.BeliefPropagationDecoder(40, early_exit=False)
    stream = torch.cuda.Stream()
    outs = []
    for H, y in zip(mats, ys):
        yd = torch.from_numpy(y).cuda()
        bits = torch.zeros((F, 3), dtype=torch.int32, device="cuda")
        ok = torch.zeros(F, dtype=torch.uint8, device="cuda")
        its = torch.zeros(F, dtype=torch.int32, device="cuda")
        stream.wait_stream(torch.cuda.current_stream())
        dec.decode_batch_dev(H, yd.data_ptr(), False, F, 1.0, bits.data_ptr(), ok.data_ptr(), its.data_ptr(), stream.cuda_stream)
        outs.append((yd, bits, ok, its))
        assert dec.live_handles() <= dec.max_handles
    torch.cuda.synchronize()
    for (yd, bits, ok, its), (wb, wok, wit) in zip(outs, want):
        got = bits.cpu().numpy().view(np.uint8)[:, :12]
        assert (ok.cpu().numpy() == wok).all() and (its.cpu().numpy() == wit).all() and (got == wb).all()
    dec.close()


@pytest.mark.gpu
def test_cxx_adaptor_100_matrices_bounded(tmp_path):
    """the C++ mirror (include/acg_ldpc_decoder.hpp): 100 distinct H through one BeliefPropagationDecoder and one
    QPADMMDecoder, from 4 host threads at once"""
    import acg_alp_ldpc_amd as A
    A.build()
    exe = str(tmp_path / "cxx_adaptor_check")
    libdir = os.path.join(ROOT, "acg_alp_ldpc_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cxx_adaptor_check.cpp"), "-o", exe, "-L" + libdir,
                           "-lacg_ldpc_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-pthread"])
    out = subprocess.run([exe, os.path.join(ROOT, "data", "H.txt"), "lru"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "lru ok" in out.stdout, out.stdout
