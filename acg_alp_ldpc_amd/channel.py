"""utils/channel.h behind the C ABI: llr_variance, gen_random_codewords, transmit (host, bit-exact)."""
import numpy as np

from ._lib import check, lib


def llr_variance(snr):  # channel.h:12
    return float(lib().acg_ldpc_llr_variance(float(snr)))


def llr(v, snr):  # channel.h:14-16
    return 2 * v / llr_variance(snr)


def gen_random_codewords(G, n, seed):
    """gen_random_codewords (channel.h:39-44) with std::mt19937(seed) — n codewords"""
    G = np.ascontiguousarray(G, dtype=np.uint8)
    out = np.zeros((n, G.shape[1]), dtype=np.uint8)
    check(lib().acg_ldpc_gen_codewords(G.ctypes.data, G.shape[0], G.shape[1], int(seed), int(n), out.ctypes.data))
    return out


def transmit_frames(codewords, snr, first_frame=0, frames=None):
    """transmit (channel.h:18-26) as exp() drives it: global frame g <- mt19937(g+1) (experiment.h:97)"""
    cw = np.ascontiguousarray(codewords, dtype=np.uint8)
    frames = cw.shape[0] if frames is None else int(frames)
    y = np.zeros((frames, cw.shape[1]), dtype=np.float64)
    check(lib().acg_ldpc_transmit_host(cw.ctypes.data, cw.shape[0], cw.shape[1], int(first_frame), frames,
                                       float(snr), y.ctypes.data))
    return y
