"""Seeded generator of synthetic regular LDPC matrices (BASELINE configs[4]: the reference has no such
generator — optimize_H.cpp:106-122 only draws random-density QC protographs)."""
import numpy as np


def regular_ldpc(m, n, dv, dc, seed=1, max_tries=200):
    """(dv, dc)-regular m x n parity-check matrix (n*dv == m*dc), no repeated edges.

    Socket (configuration-model) construction: variable sockets are permuted and dealt to check sockets;
    collisions (a variable hitting the same check twice) are repaired by swapping with random sockets."""
    if n * dv != m * dc:
        raise ValueError("n*dv must equal m*dc")
    rng = np.random.default_rng(seed)
    E = n * dv
    for _ in range(max_tries):
        vs = np.repeat(np.arange(n), dv)
        rng.shuffle(vs)
        cs = np.repeat(np.arange(m), dc)
        ok = False
        for _fix in range(200):
            key = cs.astype(np.int64) * n + vs
            order = np.argsort(key, kind="stable")
            dup = np.zeros(E, dtype=bool)
            dup[order[1:]] = key[order[1:]] == key[order[:-1]]
            bad = np.nonzero(dup)[0]
            if len(bad) == 0:
                ok = True
                break
            other = rng.integers(0, E, size=len(bad))
            vs[bad], vs[other] = vs[other].copy(), vs[bad].copy()
        if ok:
            H = np.zeros((m, n), dtype=np.uint8)
            H[cs, vs] = 1
            if (H.sum(axis=0) == dv).all() and (H.sum(axis=1) == dc).all():
                return H
    raise RuntimeError("could not build a simple regular graph")
