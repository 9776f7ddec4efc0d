"""ctypes loader for libacg_ldpc_hip.so (the C ABI in include/acg_ldpc.h).

The library is the product: there is no Python/NumPy/PyTorch decode path behind it.  If the
shared object is missing, or no HIP device is present when a decoder is created, this raises —
it never falls back to a CPU implementation (and never touches oracle/).
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
# ACG_LDPC_LIB: developer override for A/B runs of differently-built libraries (tools/ab_build.sh)
LIB_PATH = os.environ.get("ACG_LDPC_LIB") or os.path.join(HERE, "lib", "libacg_ldpc_hip.so")
CSRC = os.path.join(HERE, "csrc")


class LdpcError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [("algo", C.c_int32), ("max_iter", C.c_int32), ("alpha", C.c_double), ("mu", C.c_double),
                ("eps_stop", C.c_double), ("ms_scale", C.c_double), ("early_exit", C.c_int32),
                ("precision", C.c_int32), ("device", C.c_int32), ("lanes_per_frame", C.c_int32),
                ("engine", C.c_int32), ("fast_setup", C.c_int32), ("schedule", C.c_int32)]


class McCfg(C.Structure):
    _fields_ = [("frames", C.c_int64), ("first_frame", C.c_int64), ("snr", C.c_double), ("seed", C.c_uint64),
                ("noise", C.c_int32), ("codewords", C.c_void_p), ("n_codewords", C.c_int64)]


class McResult(C.Structure):
    _fields_ = [("correct", C.c_int64), ("pseudo", C.c_int64), ("total", C.c_int64), ("sum_hamming", C.c_int64),
                ("sum_hamming_ok", C.c_int64), ("sum_hamming_wrong", C.c_int64), ("sum_iters", C.c_int64),
                ("time_sec", C.c_double), ("kernel_ms", C.c_double)]


ALGO_BP, ALGO_MINSUM, ALGO_QPADMM = 0, 1, 2
PREC_DEFAULT, PREC_F64, PREC_F32, PREC_F16 = 0, 1, 2, 3
NOISE_DEVICE_PHILOX, NOISE_HOST_MT19937 = 0, 1
ENGINE_AUTO, ENGINE_FUSED, ENGINE_STREAMED = 0, 1, 2
SCHEDULE_FLOODING, SCHEDULE_LAYERED = 0, 1

# every symbol include/acg_ldpc.h declares: (restype, argtypes)
_vp, _i32, _i64, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
SYMBOLS = {
    "acg_ldpc_params_default": (None, [C.POINTER(Params)]),
    "acg_ldpc_last_error": (C.c_char_p, []),
    "acg_ldpc_device_available": (C.c_int, []),
    "acg_ldpc_code_from_dense": (C.c_int, [_vp, _i32, _i32, C.POINTER(_vp)]),
    "acg_ldpc_code_load_txt": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "acg_ldpc_code_save_txt": (C.c_int, [_vp, C.c_char_p]),
    "acg_ldpc_code_destroy": (None, [_vp]),
    "acg_ldpc_code_dims": (None, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "acg_ldpc_code_dense": (None, [_vp, _vp]),
    "acg_ldpc_code_admm_shape": (None, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32),
                                        C.POINTER(_f64), C.POINTER(_f64)]),
    "acg_ldpc_code_generator": (C.c_int, [_vp, _vp]),
    "acg_ldpc_code_is_codeword": (C.c_int, [_vp, _vp]),
    "acg_ldpc_decoder_create": (C.c_int, [_vp, C.POINTER(Params), C.POINTER(_vp)]),
    "acg_ldpc_decoder_destroy": (None, [_vp]),
    "acg_ldpc_decoder_name": (C.c_char_p, [_vp]),
    "acg_ldpc_decode_batch": (C.c_int, [_vp, _vp, _i64, _f64, _vp, _vp, _vp]),
    "acg_ldpc_decode_batch_f32": (C.c_int, [_vp, _vp, _i64, _f64, _vp, _vp, _vp]),
    "acg_ldpc_decode_batch_dev": (C.c_int, [_vp, _vp, _i32, _i64, _f64, _vp, _vp, _vp, _vp]),
    "acg_ldpc_decoder_sync": (C.c_int, [_vp]),
    "acg_ldpc_decoder_last_kernel_ms": (C.c_float, [_vp]),
    "acg_ldpc_decoder_layout": (None, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "acg_ldpc_decoder_describe": (_i32, [_vp, C.c_char_p, _i32]),
    "acg_ldpc_mc_run": (C.c_int, [_vp, C.POINTER(McCfg), C.POINTER(McResult)]),
    "acg_ldpc_mc_merge": (None, [C.POINTER(McResult), C.POINTER(McResult)]),
    "acg_ldpc_gen_codewords": (C.c_int, [_vp, _i32, _i32, C.c_uint32, _i64, _vp]),
    "acg_ldpc_transmit_host": (C.c_int, [_vp, _i64, _i32, _i64, _i64, _f64, _vp]),
    "acg_ldpc_llr_variance": (_f64, [_f64]),
    "acg_ldpc_awgn_dev": (C.c_int, [_vp, C.POINTER(McCfg), _vp, _vp]),
    "acg_ldpc_debug_ring_tasks": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), _vp, _vp, _i64, _vp]),
    "acg_ldpc_debug_layers": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), _vp, _i64]),
    "acg_ldpc_debug_phi": (C.c_int, [_vp, _vp, _i32, _i32]),
    "acg_ldpc_debug_bp_trace": (C.c_int, [_vp, _vp, _i32, _f64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
}

_lib = None


def build(verbose=False):
    """hipcc --offload-arch=gfx950 build of the shared library (cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=out)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LdpcError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU fallback)" % LIB_PATH)
        # PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1.  Two HIP runtimes
        # in one process cannot both own the GPU ("No HIP GPUs are available" from whichever comes second),
        # so when torch is installed it is imported FIRST: the dynamic loader then resolves this library's
        # NEEDED libamdhip64.so.7 to the copy torch already mapped (same SONAME) and the process has one runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)  # AttributeError here = header/library drift
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise LdpcError("libacg_ldpc_hip error %d: %s" % (rc, lib().acg_ldpc_last_error().decode()))
