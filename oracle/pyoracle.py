"""TEST INFRASTRUCTURE — ctypes bindings for the CPU oracle and (when built) the real reference.

`Oracle()`  -> oracle/_build/libldpc_oracle.so  (this repo's C restatement, prefix ldo_)
`Ref()`     -> oracle/_ref/libacg_ref.so        (reference headers compiled as-is, prefix acgref_)

Both expose the same Python methods so a test can run the same body against either.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package acg_alp_ldpc_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_build", "libldpc_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libacg_ref.so")

_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_longp = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


def build(force=False):
    """make -C oracle (compiles the restatement; compiles _ref only if /root/reference exists)."""
    if force or not os.path.exists(ORACLE_SO) or (
            os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "ldpc_oracle.c"))):
        subprocess.check_call(["make", "-C", HERE, "oracle"], stdout=subprocess.DEVNULL)
    subprocess.call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


def ref_available():
    return os.path.exists(REF_SO)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class _Base:
    prefix = ""
    is_ref = False

    def __init__(self, path):
        self.lib = C.CDLL(path)
        p = self.prefix
        L = self.lib

        def fn(name, res, args):
            f = getattr(L, p + name)
            f.restype = res
            f.argtypes = args
            return f

        self._read_pcm = fn("read_pcm", C.c_int, [C.c_char_p, _u8p, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_int)])
        self._save_matrix = fn("save_matrix", None if self.is_ref else C.c_int, [_u8p, C.c_int, C.c_int, C.c_char_p])
        self._llr_variance = fn("llr_variance", C.c_double, [C.c_double])
        self._llr = fn("llr", C.c_double, [C.c_double, C.c_double])
        self._transmit = fn("transmit", None, [C.c_uint32, C.c_double, _u8p, C.c_int, _f64p])
        self._get_orthogonal = fn("get_orthogonal", C.c_int, [_u8p, C.c_int, C.c_int, _u8p])
        self._gen_codewords = fn("gen_codewords", None, [_u8p, C.c_int, C.c_int, C.c_uint32, C.c_int, _u8p])
        self._is_codeword = fn("is_codeword", C.c_int, [_u8p, C.c_int, C.c_int, _u8p])
        self._bp_trace = fn("bp_trace", C.c_int, [_u8p, C.c_int, C.c_int, _f64p, C.c_double, C.c_int,
                                                  _f64p, _f64p, _f64p, _f64p])
        self._admm_shape = fn("admm_shape", None, [_u8p, C.c_int, C.c_int, _f64p])
        self._admm_matrix = fn("admm_matrix", None, [_u8p, C.c_int, C.c_int, _i32p, _i32p, _f64p, _f64p])
        self._experiment = fn("experiment", C.c_double, [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                                         _u8p, C.c_int, C.c_int, _u8p, C.c_int, C.c_double, _longp])

    # ---- text format
    def read_pcm(self, path):
        cap = max(1 << 16, os.path.getsize(path))
        buf = np.zeros(cap, dtype=np.uint8)
        m, n = C.c_int(), C.c_int()
        rc = self._read_pcm(path.encode(), buf, cap, C.byref(m), C.byref(n))
        if rc:
            raise RuntimeError("read_pcm(%s) failed rc=%d" % (path, rc))
        return buf[: m.value * n.value].reshape(m.value, n.value).copy()

    def save_matrix(self, H, path):
        H = _u8(H)
        self._save_matrix(H, H.shape[0], H.shape[1], path.encode())

    # ---- channel
    def llr_variance(self, snr):
        return self._llr_variance(float(snr))

    def llr(self, v, snr):
        return self._llr(float(v), float(snr))

    def transmit(self, seed, snr, cw):
        cw = _u8(cw)
        y = np.zeros(cw.shape[0], dtype=np.float64)
        self._transmit(int(seed), float(snr), cw, cw.shape[0], y)
        return y

    def transmit_frames(self, codewords, snr, first_seed=1):
        """frame i (0-based) gets mt19937(first_seed + i) — experiment.h:90-97 single-threaded."""
        cws = _u8(codewords)
        out = np.zeros(cws.shape, dtype=np.float64)
        for i in range(cws.shape[0]):
            out[i] = self.transmit(first_seed + i, snr, cws[i])
        return out

    # ---- GF(2)
    def get_orthogonal(self, H):
        H = _u8(H)
        m, n = H.shape
        G = np.zeros((n - m, n), dtype=np.uint8)
        ok = self._get_orthogonal(H, m, n, G)
        return (G if ok else None), bool(ok)

    def gen_codewords(self, G, seed, count):
        G = _u8(G)
        out = np.zeros((count, G.shape[1]), dtype=np.uint8)
        self._gen_codewords(G, G.shape[0], G.shape[1], int(seed), int(count), out)
        return out

    def is_codeword(self, H, c):
        H = _u8(H)
        return bool(self._is_codeword(H, H.shape[0], H.shape[1], _u8(c)))

    # ---- BP
    def bp_trace(self, H, y, snr, iters):
        H = _u8(H)
        m, n = H.shape
        E = int(H.sum())
        c2v, mag, sgn = (np.zeros(E, dtype=np.float64) for _ in range(3))
        post = np.zeros(n, dtype=np.float64)
        e = self._bp_trace(H, m, n, _f64(y), float(snr), int(iters), c2v, mag, sgn, post)
        assert e == E
        return dict(c2v=c2v, v2c_mag=mag, v2c_sgn=sgn, post=post)

    # ---- ADMM structure
    def admm_shape(self, H):
        H = _u8(H)
        out = np.zeros(5, dtype=np.float64)
        self._admm_shape(H, H.shape[0], H.shape[1], out)
        return dict(n_var=int(out[0]), n_con=int(out[1]), nnz=int(out[2]), e_min=out[3], e_max=out[4])

    def admm_matrix(self, H):
        H = _u8(H)
        s = self.admm_shape(H)
        col_ptr = np.zeros(s["n_var"] + 1, dtype=np.int32)
        con = np.zeros(s["nnz"], dtype=np.int32)
        coef = np.zeros(s["nnz"], dtype=np.float64)
        b = np.zeros(s["n_con"], dtype=np.float64)
        self._admm_matrix(H, H.shape[0], H.shape[1], col_ptr, con, coef, b)
        return col_ptr, con, coef, b

    # ---- Monte-Carlo (single thread; frame i seeded i+1)
    def experiment(self, kind, H, codewords, snr, max_iter, alpha=0.0, mu=0.0, eps=1e-5):
        H = _u8(H)
        cws = _u8(codewords)
        out = np.zeros(6, dtype=np.int64)
        kinds = {"bp": 0, "qpadmm": 1, "minsum": 2}
        t = self._experiment(kinds[kind], int(max_iter), float(alpha), float(mu), float(eps), H, H.shape[0],
                             H.shape[1], cws, cws.shape[0], float(snr), out)
        keys = ["correct", "pseudo", "total", "sum_hamming", "sum_hamming_ok", "sum_hamming_wrong"]
        r = {k: int(v) for k, v in zip(keys, out)}
        r["time_sec"] = t
        return r


class Oracle(_Base):
    prefix = "ldo_"

    def __init__(self):
        build()
        super().__init__(ORACLE_SO)
        L = self.lib
        L.ldo_bp_decode_batch.restype = C.c_double
        L.ldo_bp_decode_batch.argtypes = [_u8p, C.c_int, C.c_int, _f64p, C.c_int, C.c_double, C.c_int, C.c_int,
                                          _u8p, _u8p, _i32p]
        L.ldo_minsum_decode_batch.restype = C.c_double
        L.ldo_minsum_decode_batch.argtypes = [_u8p, C.c_int, C.c_int, _f64p, C.c_int, C.c_double, C.c_int,
                                              C.c_double, C.c_int, _u8p, _u8p, _i32p]
        L.ldo_qpadmm_decode_batch.restype = C.c_double
        L.ldo_qpadmm_decode_batch.argtypes = [_u8p, C.c_int, C.c_int, _f64p, C.c_int, C.c_double, C.c_double,
                                              C.c_double, C.c_int, C.c_double, C.c_int, _u8p, _u8p, _i32p]
        self.last_time = 0.0

    def _batch_bufs(self, H, y):
        H = _u8(H)
        y = _f64(y).reshape(-1, H.shape[1])
        F = y.shape[0]
        return H, y, F, np.zeros((F, H.shape[1]), np.uint8), np.zeros(F, np.uint8), np.zeros(F, np.int32)

    def bp_decode(self, H, y, snr, max_iter, threads=1):
        """-> bits[F,n] (zeros on failure), ok[F], iters[F]"""
        H, y, F, bits, ok, iters = self._batch_bufs(H, y)
        self.last_time = self.lib.ldo_bp_decode_batch(H, H.shape[0], H.shape[1], y, F, float(snr), int(max_iter),
                                                      int(threads), bits, ok, iters)
        return bits, ok, iters

    def minsum_decode(self, H, y, snr, max_iter, scale=1.0, threads=1):
        H, y, F, bits, ok, iters = self._batch_bufs(H, y)
        self.last_time = self.lib.ldo_minsum_decode_batch(H, H.shape[0], H.shape[1], y, F, float(snr),
                                                          int(max_iter), float(scale), int(threads), bits, ok, iters)
        return bits, ok, iters

    def qpadmm_decode(self, H, y, snr, alpha, mu, max_iter, eps=1e-5, threads=1):
        H, y, F, bits, ok, iters = self._batch_bufs(H, y)
        self.last_time = self.lib.ldo_qpadmm_decode_batch(H, H.shape[0], H.shape[1], y, F, float(snr), float(alpha),
                                                          float(mu), int(max_iter), float(eps), int(threads), bits,
                                                          ok, iters)
        return bits, ok, iters


class Ref(_Base):
    prefix = "acgref_"
    is_ref = True

    def __init__(self):
        if not os.path.exists(REF_SO):
            build()
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO)
        super().__init__(REF_SO)
        L = self.lib
        L.acgref_bp_decode_batch.restype = C.c_double
        L.acgref_bp_decode_batch.argtypes = [_u8p, C.c_int, C.c_int, _f64p, C.c_int, C.c_double, C.c_int, _u8p, _u8p]
        L.acgref_qpadmm_decode_batch.restype = C.c_double
        L.acgref_qpadmm_decode_batch.argtypes = [_u8p, C.c_int, C.c_int, _f64p, C.c_int, C.c_double, C.c_double,
                                                 C.c_double, C.c_int, C.c_double, _u8p, _u8p]
        L.acgref_reset_node_counter.restype = None
        self.last_time = 0.0

    def reset_node_counter(self):
        self.lib.acgref_reset_node_counter()

    def bp_decode(self, H, y, snr, max_iter, threads=1):
        """single-threaded by necessity (bp.h:13 static counter); iters not exposed by the reference."""
        H = _u8(H)
        y = _f64(y).reshape(-1, H.shape[1])
        F = y.shape[0]
        bits, ok = np.zeros((F, H.shape[1]), np.uint8), np.zeros(F, np.uint8)
        self.last_time = self.lib.acgref_bp_decode_batch(H, H.shape[0], H.shape[1], y, F, float(snr), int(max_iter),
                                                         bits, ok)
        return bits, ok, None

    def qpadmm_decode(self, H, y, snr, alpha, mu, max_iter, eps=1e-5, threads=1):
        H = _u8(H)
        y = _f64(y).reshape(-1, H.shape[1])
        F = y.shape[0]
        bits, ok = np.zeros((F, H.shape[1]), np.uint8), np.zeros(F, np.uint8)
        self.last_time = self.lib.acgref_qpadmm_decode_batch(H, H.shape[0], H.shape[1], y, F, float(snr),
                                                             float(alpha), float(mu), int(max_iter), float(eps),
                                                             bits, ok)
        return bits, ok, None
