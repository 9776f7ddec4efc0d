"""Monte-Carlo harness: the reference's experiment.h (ExperimentResult, multithread_experiment,
merge_exp_results) with the frame loop on the GPU.

Sharding: frames are independent, so rank r of W simulates the contiguous global
range [r*F/W, (r+1)*F/W); noise seeds derive from the GLOBAL frame index, so the union of the
shards is the same set of frames for any W.  The only cross-rank step is adding seven integers
(merge_exp_results, experiment.h:70-78) — no data-path collective.
"""
import ctypes as C
import threading
import time

import numpy as np

from . import _lib
from ._lib import McCfg, McResult, check, lib


class ExperimentResult:
    """experiment.h:49-68 (+ sum_iters / kernel_ms for the throughput report)"""
    FIELDS = ("correct", "pseudo", "total", "sum_hamming", "sum_hamming_ok", "sum_hamming_wrong", "sum_iters")

    def __init__(self, **kw):
        for f in self.FIELDS:
            setattr(self, f, int(kw.get(f, 0)))
        self.time_sec = float(kw.get("time_sec", 0.0))
        self.kernel_ms = float(kw.get("kernel_ms", 0.0))

    def FER(self):
        return (self.total - self.correct) / self.total

    def avg_time(self):
        return self.time_sec / self.total

    def mean_hamming(self):
        return self.sum_hamming / self.total

    def mean_hamming_ok(self):
        return self.sum_hamming_ok / max(1, self.correct)

    def mean_hamming_wrong(self):
        return self.sum_hamming_wrong / max(1, self.total - self.correct)

    def mean_iters(self):
        return self.sum_iters / max(1, self.total)

    def as_vector(self):
        return np.array([getattr(self, f) for f in self.FIELDS], dtype=np.int64)

    @classmethod
    def from_vector(cls, v, time_sec=0.0, kernel_ms=0.0):
        return cls(time_sec=time_sec, kernel_ms=kernel_ms, **{f: int(x) for f, x in zip(cls.FIELDS, v)})

    def __repr__(self):
        return "ExperimentResult(" + ", ".join("%s=%d" % (f, getattr(self, f)) for f in self.FIELDS) + ")"


def merge_exp_results(a, b):
    """experiment.h:70-78"""
    for f in ExperimentResult.FIELDS:
        setattr(a, f, getattr(a, f) + getattr(b, f))
    a.time_sec += b.time_sec
    a.kernel_ms += b.kernel_ms
    return a


def shard_range(frames, rank, world):
    """contiguous global frame range of `rank`"""
    lo = (frames * rank) // world
    hi = (frames * (rank + 1)) // world
    return lo, hi - lo


def run_experiment(decoder, codewords, H, snr, frames=None, first_frame=0, noise="host", seed=1):
    """multithread_experiment (experiment.h:125-139) for global frames [first_frame, first_frame+frames).

    noise="host":   bit-exact reference frames (frame g <- mt19937(g+1), libstdc++ normal_distribution);
                    frame g transmits codewords[g % len(codewords)] (the reference: one codeword per frame).
    noise="device": Philox AWGN generated inside the decode kernel (throughput runs).
    codewords=None: the all-zero codeword.
    """
    h, code = decoder.handle(H)
    cfg = McCfg()
    cw = None
    if codewords is not None:
        cw = np.ascontiguousarray(codewords, dtype=np.uint8)
        assert cw.ndim == 2 and cw.shape[1] == code.n
        cfg.codewords = cw.ctypes.data
        cfg.n_codewords = cw.shape[0]
    if frames is None:
        if cw is None:
            raise ValueError("frames required without codewords")
        frames = cw.shape[0]
    cfg.frames = int(frames)
    cfg.first_frame = int(first_frame)
    cfg.snr = float(snr)
    cfg.seed = int(seed)
    cfg.noise = _lib.NOISE_HOST_MT19937 if noise == "host" else _lib.NOISE_DEVICE_PHILOX
    res = McResult()
    check(lib().acg_ldpc_mc_run(h, C.byref(cfg), C.byref(res)))
    return ExperimentResult(**{f: getattr(res, f) for f in ExperimentResult.FIELDS}, time_sec=res.time_sec,
                            kernel_ms=res.kernel_ms)


def run_experiment_sharded(decoder, codewords, H, snr, frames, rank=0, world=1, noise="device", seed=1, group=None):
    """One rank's shard + host-side sum of the counters across ranks.

    With world > 1, torch.distributed must be initialised (gloo or nccl); the all_reduce carries
    seven integers and is control-plane only."""
    lo, cnt = shard_range(int(frames), rank, world)
    t0 = time.time()
    local = run_experiment(decoder, codewords, H, snr, frames=cnt, first_frame=lo, noise=noise, seed=seed)
    if world == 1:
        return local, local
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    v = torch.from_numpy(local.as_vector()).to(dev)
    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
    total = ExperimentResult.from_vector(v.cpu().numpy(), time_sec=time.time() - t0, kernel_ms=local.kernel_ms)
    return local, total


def run_experiment_inproc(make_decoder, codewords, H, snr, frames, devices, noise="device", seed=1):
    """All GPUs of a node from ONE process, no torch.distributed: one host thread and one decoder per device (what
    multithread_experiment does with pthreads, experiment.h:125-139), shard g = the contiguous global range
    shard_range(frames, g, len(devices)), counters merged on the host (merge_exp_results, experiment.h:70-78).

    make_decoder(device) -> a Decoder bound to that device.  Returns (per-shard results, merged total)."""
    devices = list(devices)
    W = len(devices)
    locals_ = [None] * W
    errs = []

    def work(g):
        try:
            dec = make_decoder(devices[g])
            lo, cnt = shard_range(int(frames), g, W)
            locals_[g] = run_experiment(dec, codewords, H, snr, frames=cnt, first_frame=lo, noise=noise, seed=seed)
        except Exception as e:  # noqa: BLE001 - re-raised below in the calling thread
            errs.append(e)

    t0 = time.time()
    th = [threading.Thread(target=work, args=(g,)) for g in range(W)]
    [t.start() for t in th]
    [t.join() for t in th]
    if errs:
        raise errs[0]
    total = ExperimentResult()
    for r in locals_:
        merge_exp_results(total, r)
    total.time_sec = time.time() - t0
    return locals_, total
