"""Host-side mirror of the reference's Decoder operator API (algo/algo.h:6-11).

    BeliefPropagationDecoder(max_iter)                     algo/bp.h:208-222
    QPADMMDecoder(alpha, mu, max_iter=2000, eps_stop=1e-5) algo/qp_admm.h:180-194
    MinSumDecoder(max_iter, scale)                         build-added, not in the reference

    decode(H, channel_word, snr) -> (codeword, ok)         same argument meaning as algo/algo.h:8:
        H: m x n 0/1 matrix (or a ParityCheckMatrix), channel_word: n raw channel symbols y
        (NOT LLRs), snr: Es/N0 in dB.  BP failure returns (empty array, False) like bp.h:198.
    decode_batch(H, Y, snr) -> (bits[F,n], ok[F], iters[F])
    name() -> "BP" / "QP-ADMM" / "MS"

Every call runs on the GPU through libacg_ldpc_hip.so; there is no CPU path.
"""
import collections
import contextlib
import ctypes as C
import threading

import numpy as np

from . import _lib
from ._lib import Params, check, lib
from .code import ParityCheckMatrix


class Decoder:
    _algo = None

    # The reference hands H to every decode() (algo/algo.h:8) and its optimize_H loop hands a NEW H per proposal to one
    # shared decoder (optimize_H.cpp:16-25,89-104): the analysed-graph cache is therefore bounded.  Least recently used
    # handles beyond MAX_HANDLES are destroyed (device memory, streams and events go with them).
    MAX_HANDLES = 8

    def __init__(self, max_iter, early_exit=True, precision=_lib.PREC_DEFAULT, device=-1, lanes_per_frame=0,
                 engine=_lib.ENGINE_AUTO, schedule=_lib.SCHEDULE_FLOODING, fast_setup=False, max_handles=None):
        self.max_iter = int(max_iter)
        self.early_exit = bool(early_exit)
        self.precision = precision
        self.device = device
        self.lanes_per_frame = lanes_per_frame
        self.engine = engine
        self.schedule = schedule
        self.fast_setup = bool(fast_setup)
        self.max_handles = int(max_handles) if max_handles else self.MAX_HANDLES
        self._handles = collections.OrderedDict()  # analysed-graph cache keyed on H (SURVEY §8b "Inputs"), LRU order
        self._pins = collections.Counter()         # calls in flight per key: a handle in use is never evicted
        self._lock = threading.RLock()             # (ctypes calls release the GIL: one decoder object may be used from threads)

    # -- parameters -------------------------------------------------------------------------
    def _params(self):
        p = Params()
        lib().acg_ldpc_params_default(C.byref(p))
        p.algo = self._algo
        p.max_iter = self.max_iter
        p.early_exit = 1 if self.early_exit else 0
        p.precision = self.precision
        p.device = self.device
        p.lanes_per_frame = self.lanes_per_frame
        p.engine = self.engine
        p.schedule = self.schedule
        p.fast_setup = 1 if self.fast_setup else 0
        return p

    # -- handle cache -----------------------------------------------------------------------
    def _key(self, H):
        """ParityCheckMatrix objects are keyed on identity (they are immutable); dense arrays on their CONTENT: shape + the
        packed bits themselves (5.6 KB for 160 x 280), compared for equality by the dict — a hash alone could collide and
        silently decode against the wrong graph."""
        if isinstance(H, ParityCheckMatrix):
            return ("pcm", id(H))
        a = np.ascontiguousarray(H)
        if a.ndim != 2:
            raise ValueError("H must be an m x n matrix")
        return ("dense", a.shape, np.packbits(a != 0).tobytes())

    def handle(self, H):
        """(device handle, ParityCheckMatrix) for H — valid until max_handles OTHER matrices have been used through this object;
        the decode calls below pin theirs for the duration of the call."""
        with self._lock:
            return self._lookup(self._key(H), H)

    def _lookup(self, k, H):
        ent = self._handles.get(k)
        if ent is not None:
            self._handles.move_to_end(k)
            return ent
        code = H if isinstance(H, ParityCheckMatrix) else ParityCheckMatrix(H)
        h = C.c_void_p()
        p = self._params()
        check(lib().acg_ldpc_decoder_create(code._h, C.byref(p), C.byref(h)))
        ent = (h, code)
        self._handles[k] = ent
        self._trim()
        return ent

    def _trim(self):
        if len(self._handles) <= self.max_handles:
            return
        for k in list(self._handles):           # least recently used first; never the newest, never one in use
            if len(self._handles) <= self.max_handles:
                break
            if self._pins[k] == 0 and k != next(reversed(self._handles)):
                old, _ = self._handles.pop(k)
                lib().acg_ldpc_decoder_destroy(old)   # synchronises the handle's stream first

    @contextlib.contextmanager
    def _lease(self, H):
        """the handle for H, kept out of the eviction's reach while a call uses it"""
        with self._lock:
            k = self._key(H)
            ent = self._lookup(k, H)
            self._pins[k] += 1
        try:
            yield ent
        finally:
            with self._lock:
                self._pins[k] -= 1
                self._trim()

    def live_handles(self):
        return len(self._handles)

    def close(self):
        with self._lock:
            for h, _ in self._handles.values():
                lib().acg_ldpc_decoder_destroy(h)
            self._handles = collections.OrderedDict()
            self._pins.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- reference API ----------------------------------------------------------------------
    def name(self):
        return {_lib.ALGO_BP: "BP", _lib.ALGO_QPADMM: "QP-ADMM", _lib.ALGO_MINSUM: "MS"}[self._algo]

    def decode(self, H, channel_word, snr):
        bits, ok, _ = self.decode_batch(H, np.asarray(channel_word, dtype=np.float64)[None, :], snr)
        if not ok[0] and self._algo != _lib.ALGO_QPADMM:
            return np.zeros(0, dtype=np.uint8), False  # {TCodeword(), false}, bp.h:198
        return bits[0], bool(ok[0])

    def decode_batch(self, H, Y, snr, out=None):
        """Y: frames x n channel symbols.  float64 (default): the reference's exact LLRs; a float32 array travels as
        float32 (half the PCIe bytes, acg_ldpc_decode_batch_f32).  out = (bits, ok, iters) reuses caller arrays."""
        with self._lease(H) as (h, code):
            return self._decode_batch(h, code, Y, snr, out)

    def _decode_batch(self, h, code, Y, snr, out):
        f32 = isinstance(Y, np.ndarray) and Y.dtype == np.float32
        Y = np.ascontiguousarray(Y, dtype=np.float32 if f32 else np.float64)
        if Y.ndim != 2 or Y.shape[1] != code.n:
            raise ValueError("Y must be frames x n")
        F = Y.shape[0]
        if out is not None:
            bits, ok, iters = out
            assert bits.shape == (F, code.n) and bits.dtype == np.uint8 and bits.flags.c_contiguous
            assert ok.shape == (F,) and ok.dtype == np.uint8 and iters.shape == (F,) and iters.dtype == np.int32
        else:
            bits = np.empty((F, code.n), dtype=np.uint8)
            ok = np.empty(F, dtype=np.uint8)
            iters = np.empty(F, dtype=np.int32)
        fn = lib().acg_ldpc_decode_batch_f32 if f32 else lib().acg_ldpc_decode_batch
        check(fn(h, Y.ctypes.data, F, float(snr), bits.ctypes.data, ok.ctypes.data, iters.ctypes.data))
        return bits, ok, iters

    def decode_batch_dev(self, H, y_ptr, y_is_f64, frames, snr, bits_ptr, ok_ptr, iters_ptr=None, stream=None):
        """device pointers (ints); asynchronous on `stream` (None = the decoder's own stream)"""
        with self._lease(H) as (h, _):
            check(lib().acg_ldpc_decode_batch_dev(h, y_ptr, 1 if y_is_f64 else 0, int(frames), float(snr), bits_ptr,
                                                  ok_ptr, iters_ptr, stream))

    def sync(self, H):
        h, _ = self.handle(H)
        check(lib().acg_ldpc_decoder_sync(h))

    def last_kernel_ms(self, H):
        h, _ = self.handle(H)
        return float(lib().acg_ldpc_decoder_last_kernel_ms(h))

    def describe(self, H):
        """one line naming the engine / kernel instance / launch shape of the handle for H"""
        h, _ = self.handle(H)
        buf = C.create_string_buffer(1024)
        lib().acg_ldpc_decoder_describe(h, buf, 1024)
        return buf.value.decode()

    def layout(self, H):
        h, _ = self.handle(H)
        a, b, c, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        lib().acg_ldpc_decoder_layout(h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return dict(lds_bytes_per_frame=a.value, lanes_per_frame=b.value, frames_per_block=c.value,
                    grid_blocks=d.value)


class BeliefPropagationDecoder(Decoder):
    """algo/bp.h:208-222 — flooding sum-product in the phi domain, early exit at the first zero syndrome."""
    _algo = _lib.ALGO_BP

    def __init__(self, max_iter, **kw):
        super().__init__(max_iter, **kw)


class MinSumDecoder(Decoder):
    """Build-added normalised min-sum (north_star).  Not in the reference: parity unpinned.
    schedule=SCHEDULE_LAYERED: layered (row-block sequential) schedule — about half the sweeps for the same FER."""
    _algo = _lib.ALGO_MINSUM

    def __init__(self, max_iter, scale=1.0, **kw):
        super().__init__(max_iter, **kw)
        self.scale = float(scale)

    def _params(self):
        p = super()._params()
        p.ms_scale = self.scale
        return p


class QPADMMDecoder(Decoder):
    """algo/qp_admm.h:180-194 (same defaults: max_iter=2000, eps_stop=1e-5)"""
    _algo = _lib.ALGO_QPADMM

    def __init__(self, alpha, mu, max_iter=2000, eps_stop=1e-5, **kw):
        super().__init__(max_iter, **kw)
        self.alpha, self.mu, self.eps_stop = float(alpha), float(mu), float(eps_stop)

    def _params(self):
        p = super()._params()
        p.alpha, p.mu, p.eps_stop = self.alpha, self.mu, self.eps_stop
        return p
