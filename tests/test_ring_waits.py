"""The riskiest construct of the streamed engine: bp_streamed_ring_kernel waits for "task i has landed" with a COUNTED
`s_waitcnt vmcnt(N_i)` whose N_i comes from a host-built table (csrc/code.cpp ring_tasks_build).  vmcnt retires vector-memory
operations in issue order, so the wait is safe iff N_i <= the number of operations CERTAINLY issued behind the last load of
task i at the moment of the wait; an over-count would read an LDS slot before its DMA has landed (first visible as a rare
soft-value error).  This test replays the kernel's issue order per wavefront — prologue loads of R - 1 tasks, then per task:
loads of task i + R - 1, the wait, the stores of task i — for regular and very irregular degree sequences, in the normal
sweeps and in the syndrome-only sweep (no stores), and checks every wait.  CPU only (the tables are host code)."""
import ctypes as C
import os

import numpy as np
import pytest

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")


def _tables(A, Hm):
    H = A.ParityCheckMatrix(Hm)
    nc, nv = C.c_int32(), C.c_int32()
    consts = np.zeros(4, dtype=np.int32)
    assert A.lib().acg_ldpc_debug_ring_tasks(H._h, C.byref(nc), C.byref(nv), None, None, 0, consts.ctypes.data) == 0
    ct = np.zeros((max(nc.value, 1), 4), dtype=np.int32)
    vt = np.zeros((max(nv.value, 1), 4), dtype=np.int32)
    cap = max(ct.size, vt.size)
    assert A.lib().acg_ldpc_debug_ring_tasks(H._h, C.byref(nc), C.byref(nv), ct.ctypes.data, vt.ctypes.data, cap, None) == 0
    return ct[:nc.value], vt[:nv.value], [int(x) for x in consts]


def _replay(tasks, W, R, var, stores_issued=True):
    """-> list of (task index, N used by the kernel, operations certainly issued behind the task's last load at its wait)"""
    out = []
    for w in range(W):
        seq = list(range(w, len(tasks), W))
        issued = 0                 # certain operations issued so far by this wavefront in this sweep
        last_load = {}             # task -> value of `issued` right after its last load
        def issue(i):
            nonlocal issued
            lines = int(tasks[seq[i]][3]) & 0xFF
            issued += (lines + 3) // 4 + (1 if var else 0)     # one LDS-DMA per four lines (+ the LLR lines of a variable task)
            last_load[i] = issued
        for i in range(min(R - 1, len(seq))):
            issue(i)
        for i in range(len(seq)):
            if i + R - 1 < len(seq):
                issue(i + R - 1)
            pk = int(tasks[seq[i]][3])
            n = ((pk >> 8) & 0xFF) if stores_issued else ((pk >> 16) & 0xFF)
            n_used = min((n >> 2) * 4, 60)                     # ring_wait_vmcnt rounds down to a multiple of 4
            out.append((seq[i], n, n_used, issued - last_load[i]))
            if stores_issued:
                issued += pk & 0xFF                            # one message store per line (the hard-decision byte is predicated: not counted)
    return out


def _ragged(m, n, rng, max_c=16, max_v=12):
    H = np.zeros((m, n), dtype=np.uint8)
    for v in range(n):
        d = int(rng.integers(0, max_v + 1))
        H[rng.choice(m, size=min(d, m), replace=False), v] = 1
    for c in range(m):
        on = np.nonzero(H[c])[0]
        if len(on) > max_c:
            H[c, on[max_c:]] = 0
    return H


def test_counted_waits_never_exceed_the_operations_issued(matrices):
    import acg_alp_ldpc_amd as A
    rng = np.random.default_rng(5)
    cases = [matrices["H"], matrices["H05"], matrices["optimalH"], A.regular_ldpc(300, 600, 3, 6, seed=2)]
    cases += [_ragged(37, 90, rng), _ragged(64, 64, rng, 16, 12), _ragged(5, 200, rng, 16, 3), _ragged(200, 9, rng, 4, 12),
              _ragged(120, 300, rng, 7, 2)]
    total = 0
    for Hm in cases:
        ct, vt, (W, R, SL, VL) = _tables(A, Hm)
        assert (W, R) == (4, 3) and VL == SL - 4
        assert ((ct[:, 3] & 0xFF) <= SL).all() and ((vt[:, 3] & 0xFF) <= VL).all() and (vt[:, 1] <= 4).all()
        # every check / variable in exactly one task, in order
        assert ct[:, 1].sum() == Hm.shape[0] and vt[:, 1].sum() == Hm.shape[1]
        assert (np.cumsum(np.r_[0, ct[:-1, 1]]) == ct[:, 0]).all() and (np.cumsum(np.r_[0, vt[:-1, 1]]) == vt[:, 0]).all()
        for tasks, var in ((ct, False), (vt, True)):
            for stores in ((True, False) if not var else (True,)):    # the check sweep also runs syndrome-only (no stores)
                for ti, n, n_used, behind in _replay(tasks, W, R, var, stores):
                    assert n_used <= behind, (Hm.shape, var, stores, ti, n, behind)          # SAFE: never reads a slot early
                    assert n == min(behind, 63), (Hm.shape, var, stores, ti, n, behind)      # and exact: never waits longer than needed
                    total += 1
    assert total > 500
