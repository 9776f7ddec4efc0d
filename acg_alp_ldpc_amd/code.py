"""Parity-check matrix handle: the reference's TMatrix + read_pcm/save_matrix/GetOrtogonal/IsCodeword
(utils/parse_data.h, utils/codeword.h) behind the C ABI."""
import ctypes as C

import numpy as np

from ._lib import check, lib


class ParityCheckMatrix:
    """Analysed once (Tanner graph CSR, LDS layouts, QP-ADMM groups); immutable afterwards."""

    def __init__(self, H=None, path=None):
        L = lib()
        self._h = C.c_void_p()
        if path is not None:
            check(L.acg_ldpc_code_load_txt(str(path).encode(), C.byref(self._h)))  # read_pcm, parse_data.h:6
        else:
            H = np.ascontiguousarray(H, dtype=np.uint8)
            if H.ndim != 2:
                raise ValueError("H must be 2-D")
            check(L.acg_ldpc_code_from_dense(H.ctypes.data, H.shape[0], H.shape[1], C.byref(self._h)))
        m, n, e = C.c_int32(), C.c_int32(), C.c_int32()
        L.acg_ldpc_code_dims(self._h, C.byref(m), C.byref(n), C.byref(e))
        self.m, self.n, self.E = m.value, n.value, e.value

    @classmethod
    def read_pcm(cls, path):
        return cls(path=path)

    def save_matrix(self, path):  # parse_data.h:44
        check(lib().acg_ldpc_code_save_txt(self._h, str(path).encode()))

    def dense(self):
        out = np.zeros((self.m, self.n), dtype=np.uint8)
        lib().acg_ldpc_code_dense(self._h, out.ctypes.data)
        return out

    def admm_shape(self):
        nv, nc, nz = C.c_int32(), C.c_int32(), C.c_int32()
        emin, emax = C.c_double(), C.c_double()
        lib().acg_ldpc_code_admm_shape(self._h, C.byref(nv), C.byref(nc), C.byref(nz), C.byref(emin), C.byref(emax))
        return dict(n_var=nv.value, n_con=nc.value, nnz=nz.value, e_min=emin.value, e_max=emax.value)

    def get_orthogonal(self):
        """GetOrtogonal (codeword.h:97-128) -> (G, ok)"""
        if self.n <= self.m:
            return None, False
        G = np.zeros((self.n - self.m, self.n), dtype=np.uint8)
        rc = lib().acg_ldpc_code_generator(self._h, G.ctypes.data)
        if rc == 1:
            return None, False
        check(rc)
        return G, True

    def layers(self):
        """layers of the layered min-sum schedule (SCHEDULE_LAYERED): (lanes per frame G, circulant size Z or 0,
        [n_layers, G] check ids in processing order, -1 = empty lane)"""
        G, nl, Z = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib().acg_ldpc_debug_layers(self._h, C.byref(G), C.byref(nl), C.byref(Z), None, 0))
        chk = np.full((nl.value, G.value), -1, dtype=np.int32)
        check(lib().acg_ldpc_debug_layers(self._h, C.byref(G), C.byref(nl), C.byref(Z), chk.ctypes.data, chk.size))
        return G.value, Z.value, chk

    def is_codeword(self, bits):
        b = np.ascontiguousarray(bits, dtype=np.uint8)
        assert b.shape[-1] == self.n
        return bool(lib().acg_ldpc_code_is_codeword(self._h, b.ctypes.data))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().acg_ldpc_code_destroy(self._h)
                self._h = None
        except Exception:
            pass
