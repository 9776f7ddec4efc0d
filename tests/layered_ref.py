"""numpy restatement (fp32, vectorised over frames) of the layered normalised min-sum of csrc/bp_layered.hip — the repo's OWN
restatement: the layered schedule is not in the reference (algo/bp.h:183-199 floods) and neither is min-sum, so
what this checks is that the kernel does what its description says, operation for operation; parity with the reference is
FER-level only (tests/test_layered.py)."""
import numpy as np


def layered_minsum(Hm, layers, y, snr, max_iter, scale, msg_dtype=np.float32):
    """Hm: m x n 0/1; layers: [n_layers, G] check ids (-1 = none) in processing order; y: frames x n float64 symbols.
    msg_dtype: storage type of the check-to-variable messages (np.float16 = precision PREC_F16: the scaled minima are rounded
    to half precision once per check; the posteriors stay fp32 and add / subtract exactly the rounded message).
    -> bits [F, n] uint8 (zeros for failed frames), ok [F] uint8, iters [F] int32"""
    Hm = np.asarray(Hm)
    F, n = y.shape
    var = 10.0 ** (-(snr / 10.0)) / 2.0
    P = (2.0 * y.astype(np.float64) / var).astype(np.float32)          # channel.h:14-16, rounded to the kernel's fp32
    edges = [np.nonzero(Hm[c])[0] for c in range(Hm.shape[0])]         # variables ascending
    R = [np.zeros((F, len(e)), dtype=np.float32) for e in edges]
    scale = np.float32(scale)
    done = np.zeros(F, dtype=bool)
    bits = np.zeros((F, n), dtype=np.uint8)
    ok = np.zeros(F, dtype=np.uint8)
    iters = np.full(F, max_iter, dtype=np.int32)
    for it in range(1, max_iter + 1):
        live = ~done
        if not live.any():
            break
        loud = np.zeros(F, dtype=bool)
        for layer in layers:
            for c in layer:
                if c < 0:
                    continue
                v = edges[c]
                p = P[:, v]
                q = p - R[c]
                a = np.abs(q)
                srt = np.sort(a, axis=1)
                m1, m2 = srt[:, 0], (srt[:, 1] if a.shape[1] > 1 else np.full(F, np.inf, np.float32))
                # the kernel forms scale * min and rounds it to the storage type in ONE step (v_fma_mixlo_f16 for fp16): the
                # product of two fp32 numbers is exact in float64, so rounding that once is the same thing
                with np.errstate(over="ignore"):
                    m1s = (np.float64(scale) * m1.astype(np.float64)).astype(msg_dtype).astype(np.float32)
                    m2s = (np.float64(scale) * m2.astype(np.float64)).astype(msg_dtype).astype(np.float32)
                if a.shape[1] == 1:       # a one-variable check: its message saturates instead of being infinite
                    m2s = np.full(F, np.float32(59968.0), dtype=np.float32)
                mag = np.where(a == m1[:, None], m2s[:, None], m1s[:, None]).astype(np.float32)
                sq = np.signbit(q)
                S = np.logical_xor.reduce(sq, axis=1)
                neg = S[:, None] ^ sq
                rn = np.where(neg, -mag, mag).astype(np.float32)
                pn = (q + rn).astype(np.float32)
                parity = np.logical_xor.reduce(np.signbit(p), axis=1)
                loud |= parity | (np.signbit(pn) != np.signbit(p)).any(axis=1)
                P[np.ix_(live, v)] = pn[live]
                R[c][live] = rn[live]
        newly = live & ~loud
        bits[newly] = np.signbit(P[newly]).astype(np.uint8)
        ok[newly] = 1
        iters[newly] = it
        done |= newly
    rest = ~done
    if rest.any() and max_iter > 0:     # out of iterations without a quiet round: one explicit syndrome pass
        hb = np.signbit(P[rest]).astype(np.uint8)
        good = ((hb @ Hm.T.astype(np.int64)) % 2 == 0).all(axis=1)
        idx = np.nonzero(rest)[0][good]
        bits[idx] = hb[good]
        ok[idx] = 1
    return bits, ok, iters


def layered_sumproduct(Hm, layers, y, snr, max_iter, sat=83.25 / 1.4426950408889634):
    """float64 restatement of the SUM-PRODUCT variant of the layered kernel (the reference's check rule, bp.h:49-57, with
    phi(x) = -log(tanh(x/2)) evaluated exactly; messages saturate at `sat` natural units like the kernel's).  The kernel works in
    fp32 with its own phi, so agreement is a RATE (tests/test_layered.py), not word for word."""
    Hm = np.asarray(Hm)
    F, n = y.shape
    var = 10.0 ** (-(snr / 10.0)) / 2.0
    P = 2.0 * y.astype(np.float64) / var
    edges = [np.nonzero(Hm[c])[0] for c in range(Hm.shape[0])]
    R = [np.zeros((F, len(e))) for e in edges]
    done = np.zeros(F, dtype=bool)
    bits = np.zeros((F, n), dtype=np.uint8)
    ok = np.zeros(F, dtype=np.uint8)
    iters = np.full(F, max_iter, dtype=np.int32)

    def phi(x):
        with np.errstate(divide="ignore", over="ignore", invalid="ignore"):
            return np.where(x >= 45.747713916956390, 0.0, -np.log(np.tanh(0.5 * x)))
    for it in range(1, max_iter + 1):
        live = ~done
        if not live.any():
            break
        loud = np.zeros(F, dtype=bool)
        for layer in layers:
            for c in layer:
                if c < 0:
                    continue
                v = edges[c]
                p = P[:, v]
                q = p - R[c]
                mag = phi(np.abs(q))
                # exclude-self sums formed directly (never total - own: an infinite term would turn into NaN), bp.h:50-55
                others = np.stack([np.delete(mag, j, axis=1).sum(axis=1) for j in range(mag.shape[1])], axis=1)
                out = np.minimum(phi(others), sat)
                sq = np.signbit(q)
                neg = np.logical_xor.reduce(sq, axis=1)[:, None] ^ sq
                rn = np.where(neg, -out, out)
                pn = q + rn
                parity = np.logical_xor.reduce(np.signbit(p), axis=1)
                loud |= parity | (np.signbit(pn) != np.signbit(p)).any(axis=1)
                P[np.ix_(live, v)] = pn[live]
                R[c][live] = rn[live]
        newly = live & ~loud
        bits[newly] = np.signbit(P[newly]).astype(np.uint8)
        ok[newly] = 1
        iters[newly] = it
        done |= newly
    rest = ~done
    if rest.any() and max_iter > 0:
        hb = np.signbit(P[rest]).astype(np.uint8)
        good = ((hb @ Hm.T.astype(np.int64)) % 2 == 0).all(axis=1)
        idx = np.nonzero(rest)[0][good]
        bits[idx] = hb[good]
        ok[idx] = 1
    return bits, ok, iters
