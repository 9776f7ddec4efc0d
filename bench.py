#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its config: decoded frames/s for data/H05.txt, 50-iteration
sum-product BP, 1M synthetic AWGN frames per GPU (BASELINE configs[1]), 1/2/4/8 MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W          # one process per GPU
    python bench.py --inproc N --steps K --warmup W                     # one process, N GPUs, no torch.distributed

A "step" is one pass of the hot path (acg_ldpc_decode_batch_dev) over one resident batch of `--frames` noisy
frames per GPU.  The batch (channel symbols, fp32) is generated on the device before the timed region; inputs and
outputs stay in HBM.  Frames are independent, so shard s owns global frames [s*F, (s+1)*F) — weak scaling, no
data-path collective.  Across processes the only traffic is the barrier, a MAX of the wall time and
a SUM of four counters, carried by gloo on CPU tensors (no RCCL anywhere: north_star).

Headline `value`: FIXED WORK — every frame runs all 50 flooding iterations (output latched at its first zero
syndrome), nothing is skipped.  The reference's own stopping rule (bp.h:195-196) is timed under "early_exit".

OUTPUT.  The LAST stdout line is a compact JSON object (< 4 KB: `compact_line`) with the headline, its roofline, the CPU
baselines and one {value, frac, bound} triple per side leg — what the driver's record keeps.  The full object (every leg
with its layout, counters and notes, ~25 KB) is written to ./bench_detail.json and printed on an EARLIER stdout line
prefixed "BENCH_DETAIL ".  Nothing is printed after the compact line.

Side legs (N = 1, or any N with --all; N > 1 runs headline + early exit + Monte-Carlo only):
  "streamed"      the HBM-resident engine north_star sketches (messages [edge][frame] in HBM): real HBM roofline
  "minsum_0.75"   north_star's named variant (not in the reference: parity unpinned)
  "configs[2]"    H05 QP-ADMM(1.95, 0.5) fp64, 100 sweeps, 1M frames — with its own roofline (LDS array)
  "configs[4]"    synthetic (3,6)-regular 5000 x 10000, min-sum 50 iterations, 32768 frames per GPU: fused and streamed
  "monte_carlo"   acg_ldpc_mc_run: AWGN generated in the kernel + classification + D2H of the counters — the
                  metric exactly as SURVEY §8(d) words it
  "cpu_baseline", "cpu_baseline_qpadmm"   the reference's bp.h / qp_admm.h on this host's cores
  "pmc"           rocprofv3 counter passes taken by THIS run (child processes) that the roofline fractions use

Roofline fractions are utilisations of the resource that binds each kernel (<= 1): VALU issue for the LDS-resident
BP kernels, the LDS array for QP-ADMM, HBM for the streamed engine.  The SURVEY §8(d) streamed-model figure of an
LDS-resident kernel is kept as `streamed_equiv_frac`; it is NOT a utilisation and may exceed 1.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_CU, N_SIMD = 256, 1024   # 256 CUs x 4 SIMD-32
NOMINAL_CLK_HZ = 2.4e9
# VALU issue prices, cycles per wave64 instruction per SIMD with several waves resident (tools/microbench/valu_rates.hip,
# output committed as profiles/r02_valu_rates.txt; MI355X_MICROARCH.md cycle-constants table: fma 2 on the SIMD-32)
CYC_VALU = 2.0
CYC_TRANS = 8.0            # v_exp_f32 / v_log_f32 alone (quarter rate); the mul+exp PAIR measures 10-12
CYC_VALU_F64 = 4.0         # fp64 add / mul / fma / max: half rate

PROBE_ITEMS = ("bp_fused", "bp_exit", "bp_mc", "bp_streamed", "ms_streamed", "ms_layered", "ms_layered_f16", "bp_layered", "qpadmm", "c5_block_ms", "c5_pair_f16_ms", "c5_streamed_ms")
PMC_PASSES = (("fetch", ["FETCH_SIZE"]),
              ("write", ["WRITE_SIZE", "GRBM_GUI_ACTIVE"]),
              ("sq", ["SQ_INSTS_VALU", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE",
                      "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES"]),
              # optional (a failure of this pass only drops the fp64 split of the QP-ADMM VALU estimate)
              ("f64", ["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"]))
PMC_OPTIONAL = ("f64",)


def bp_bytes_per_frame(n, E, iters, b=4, b_in=4):
    """B_bp = n*b_in + I*(4E + n)*b + ceil(n/8)"""
    return n * b_in + iters * (4 * E + n) * b + (n + 7) // 8


def admm_bytes_per_frame(n, n_con, n_var, iters, b=8, b_in=4):
    """B_admm = n*b_in + I*(5C + 3V)*b + ceil(n/8)"""
    return n * b_in + iters * (5 * n_con + 3 * n_var) * b + (n + 7) // 8


def csrc_sha():
    """hash of the kernel sources: a committed counter file is only used for the code it was taken from"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "acg_alp_ldpc_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".inc", ".hpp", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


# ----------------------------------------------------------------------------------------------------------------
# CPU baselines: the reference's own decoders (oracle/_ref, compiled from /root/reference in the build container) or,
# if that .so did not travel, the oracle port — one single-threaded process per host core, disjoint frame ranges
# (BP's global node counter forbids threads), on a bounded sample of the same workload.
def cpu_baseline_worker(args):
    kind, algo, Hm, y, snr, max_iter, alpha, mu = args
    if kind == "reference":
        from oracle.pyoracle import Ref
        d = Ref()
    else:
        from oracle.pyoracle import Oracle
        d = Oracle()
    t0, c0 = time.time(), time.process_time()
    if algo == "bp":
        _, ok, _ = d.bp_decode(Hm, y, snr, max_iter)
    else:
        _, ok, _ = d.qpadmm_decode(Hm, y, snr, alpha, mu, max_iter, 1e-5)
    return time.time() - t0, int(ok.sum()), time.process_time() - c0


def _spin_worker(barrier, seconds, q):
    """wait until every spinner is up, then busy-loop for `seconds` of wall time -> CPU seconds this process was given"""
    barrier.wait()
    t_end, c0 = time.time() + seconds, time.process_time()
    x = 0
    while time.time() < t_end:
        for _ in range(20000):
            x += 1
    q.put(time.process_time() - c0)


def spin_parallelism(ctx, n, seconds=2.0):
    """n single-threaded spinners started together -> (CPU seconds received in total, wall seconds)"""
    barrier, q = ctx.Barrier(n + 1), ctx.Queue()
    ps = [ctx.Process(target=_spin_worker, args=(barrier, seconds, q), daemon=True) for _ in range(n)]
    [p.start() for p in ps]
    barrier.wait(timeout=120)
    t0 = time.time()
    cpu = [q.get(timeout=seconds * 20 + 60) for _ in range(n)]
    wall = time.time() - t0
    [p.join() for p in ps]
    return sum(cpu), wall


def cgroup_cpu_quota():
    """CPUs the cgroup lets this process use at once (cpu.max / cfs quota), or None if unlimited / unreadable"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return q / per
    except Exception:
        pass
    return None


def usable_cores(pool_ctx=None):
    """How many cores this process can keep busy: os.cpu_count() says what the MACHINE has (256 on the GPU box) — the
    affinity mask and the cgroup quota say what the lease gets, and a one-second spin test with one process per candidate
    core measures it (CPU seconds received / wall seconds), which also catches limits neither file shows."""
    import math
    visible = os.cpu_count() or 1
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = visible
    quota = cgroup_cpu_quota()
    cap = aff if quota is None else max(1, min(aff, int(math.floor(quota + 0.5))))
    info = {"cores_visible": visible, "cores_affinity": aff, "cgroup_quota": quota}
    measured = None
    if pool_ctx is not None and cap > 1:
        n = min(cap, 64)
        try:
            spin_parallelism(pool_ctx, n, 0.5)     # the first burst after an idle period under-reads on a VM: discard it
            cpu, wall = max((spin_parallelism(pool_ctx, n) for _ in range(2)), key=lambda r: r[0] / r[1])
            measured = cpu / wall
            info["spin_test"] = {"processes": n, "cpu_seconds": cpu, "wall_seconds": wall, "parallelism": measured}
            if measured > 0.85 * n:
                measured = None      # no limit found below the test's own size: keep the file-based cap
        except Exception as e:       # noqa: BLE001 - the file-based cap stands then
            info["spin_test"] = {"error": repr(e)}
    used = cap if measured is None else max(1, min(cap, int(measured + 0.5)))
    info["cores_used"] = used
    return used, info


def cpu_baseline(algo, Hm, cws, snr, max_iter, target_s, alpha=0.0, mu=0.0, cores=None, core_info=None):
    """sample sized by a short calibration so the timed part is about target_s seconds of wall time on every core"""
    import multiprocessing as mp
    import resource
    import numpy as np
    from oracle.pyoracle import Oracle, ref_available
    o = Oracle()
    kind = "reference" if ref_available() else "port"
    ctx = mp.get_context("spawn")
    if cores is None:
        cores, core_info = usable_cores(ctx)
    cal = 32 if algo == "bp" else 8   # BP frames take 1 ... 50 sweeps at -2 dB: a longer calibration sample
    ycal = o.transmit_frames(cws[np.arange(cal) % len(cws)], snr, first_seed=1)
    ru0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    with ctx.Pool(cores) as pool:
        # start the workers, load the .so, and calibrate with every core busy (the rate per core depends on that)
        pool.map(cpu_baseline_worker, [(kind, algo, Hm, ycal, snr, max_iter, alpha, mu)] * cores, chunksize=1)
        res = pool.map(cpu_baseline_worker, [(kind, algo, Hm, ycal, snr, max_iter, alpha, mu)] * cores, chunksize=1)
        per_frame = max(r[0] for r in res) / cal
        per_proc = int(min(20000, max(cal, target_s / max(per_frame, 1e-6))))
        total = per_proc * cores
        y = o.transmit_frames(cws[np.arange(total) % len(cws)], snr, first_seed=1)
        chunks = [(kind, algo, Hm, y[i * per_proc:(i + 1) * per_proc], snr, max_iter, alpha, mu) for i in range(cores)]
        t0 = time.time()
        res = pool.map(cpu_baseline_worker, chunks, chunksize=1)
        wall = time.time() - t0
        pool.close()
        pool.join()
    ru1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    cpu_s = sum(r[2] for r in res)
    what = ("BeliefPropagationDecoder(%d) (algo/bp.h:208-222)" % max_iter if algo == "bp" else
            "QPADMMDecoder(%g, %g, %d, 1e-5) (algo/qp_admm.h:104-194)" % (alpha, mu, max_iter))
    survey_1t = SURVEY_SINGLE_THREAD.get((algo, snr))
    out = {
        "value": total / wall, "unit": "frames/s", "cores": cores, "kind": kind,
        "sample": "%d frames (%d per process, %d single-threaded processes) of the same H05/AWGN workload at %.1f dB, %s %s "
                  "with its own stopping rule; pool wall %.1f s (slowest process %.1f s)"
                  % (total, per_proc, cores, snr, "the reference's" if kind == "reference" else "oracle port of",
                     what, wall, max(r[0] for r in res)),
        "decoded_ok": sum(r[1] for r in res), "frames": total,
        "cores_visible": (core_info or {}).get("cores_visible"), "cores_used": cores,
        "cpu_seconds_timed_sample": cpu_s,                      # summed process CPU time of the timed sample alone
        "busy_cores_mean": cpu_s / wall,                        # CPU seconds received per wall second
        "child_cpu_seconds_whole_leg": (ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime),
        "frames_per_s_per_core": total / wall / cores,
        "frames_per_cpu_second": total / cpu_s if cpu_s > 0 else None,
        "survey_single_thread_frames_per_s": survey_1t,        # BASELINE.md / SURVEY §6: Xeon 2.10 GHz, 1 thread
        "core_detection": core_info,
    }
    return out


# BASELINE.md §2 / SURVEY §6 "measured here" (survey container, Xeon 2.10 GHz, one thread, 1000 frames)
SURVEY_SINGLE_THREAD = {("bp", -2.0): 148.0, ("qpadmm", -2.0): 532.0, ("bp", 2.0): 538.0, ("qpadmm", 2.0): 1214.0}


# ----------------------------------------------------------------------------------------------------------------
class Dev:
    """one GPU driven by this process"""

    def __init__(self, idx, shard, cpu=False):
        import torch
        self.idx, self.shard, self.stream = idx, shard, None
        if not cpu:
            with torch.cuda.device(idx):
                self.stream = torch.cuda.Stream(device=idx)   # a dedicated (non-null) HIP stream: kernels and timing events


class Rig:
    """The set of GPUs of this job: `world` processes (torch.distributed, gloo control plane) x len(devs) GPUs each."""

    def __init__(self, a, cpu=False):
        """cpu=True: no GPU is touched (tests/test_bench_line.py rehearses the launchers' control plane — shard order, the
        gloo reductions, the per-shard gather — with an oracle-backed shard function)"""
        import torch
        self.torch = torch
        self.cpu = cpu
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        ndev = max(1, 0 if cpu else torch.cuda.device_count())
        if a.inproc:
            if self.world > 1:
                raise SystemExit("--inproc runs in ONE process; do not combine it with torch.distributed.run")
            if a.inproc > ndev and not cpu and not os.environ.get("ACG_BENCH_SHARE_GPU"):
                raise SystemExit("--inproc %d but only %d GPUs visible" % (a.inproc, ndev))
            self.devs = [Dev(i % ndev, i, cpu) for i in range(a.inproc)]   # (ACG_BENCH_SHARE_GPU: rehearsal on a 1-GPU box)
            self.launcher = "inproc (one process, %d GPUs, asynchronous launches from one host thread)" % a.inproc
        else:
            if self.world != a.gpus and self.world > 1:
                raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, self.world))
            if a.gpus > 1 and self.world == 1:
                raise SystemExit("launch N>1 with torch.distributed.run (one process per GPU) or use --inproc N")
            self.devs = [Dev(local_rank % ndev, self.rank, cpu)]
            self.launcher = "torch.distributed.run, one process per GPU" if self.world > 1 else "single process"
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            # control plane only (barrier, MAX of one double, SUM of four integers) -> gloo on CPU tensors.
            # ACG_BENCH_BACKEND=nccl is kept for comparison; nothing on the data path ever uses a collective.
            self.backend = os.environ.get("ACG_BENCH_BACKEND", "gloo")
            if dist.is_initialized():      # (a test harness that set the group up itself)
                pass
            elif self.backend == "nccl":
                torch.cuda.set_device(self.devs[0].idx)
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.devs[0].idx))
            else:
                dist.init_process_group(self.backend)
            self.dist = dist
        self.nshards = self.world * len(self.devs)

    def sync_local(self):
        if self.cpu:
            return
        for d in self.devs:
            self.torch.cuda.synchronize(d.idx)

    def barrier(self):
        self.sync_local()
        if self.dist:
            self.dist.barrier()
        self.sync_local()

    def _red(self, vals, op, dtype):
        if not self.dist:
            return vals
        t = self.torch.tensor(vals, dtype=dtype, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=op)
        return t.tolist()

    def max_time(self, dt):
        return self._red([dt], self.dist.ReduceOp.MAX, self.torch.float64)[0] if self.dist else dt

    def sum_ints(self, v):
        return [int(x) for x in (self._red(list(v), self.dist.ReduceOp.SUM, self.torch.int64) if self.dist else v)]

    def gather_floats(self, vals):
        """vals: this process's per-device figures (shard order) -> the figures of every shard of the job, shard order"""
        if not self.dist:
            return [float(x) for x in vals]
        t = self.torch.zeros(self.nshards, dtype=self.torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        for d, x in zip(self.devs, vals):
            t[d.shard] = float(x)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)   # every shard is written by exactly one rank
        return t.tolist()

    def timed(self, step, steps, warmup):
        """step(dev) enqueues ONE step on dev.stream.
        -> (seconds for `steps` steps between the two barriers, MAX over ranks; mean kernel ms over this process's launches;
            {"per_shard_ms_per_step": wall of every shard from the first barrier to ITS OWN last step done, / steps;
             "per_shard_kernel_ms": mean HIP-event time of every shard's launches; "kernel_ms_per_launch": shard 0's})"""
        torch = self.torch
        for _ in range(warmup):
            for d in self.devs:
                step(d)
        ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
              for _ in self.devs]
        self.barrier()
        t0 = time.perf_counter()
        for s in range(steps):
            for i, d in enumerate(self.devs):
                with torch.cuda.device(d.idx):
                    ev[i][s][0].record(d.stream)   # HIP events on the stream the kernel is launched on
                    step(d)
                    ev[i][s][1].record(d.stream)
        own = []
        for d in self.devs:                        # a straggling shard shows here, before the barrier levels the ranks
            torch.cuda.synchronize(d.idx)
            own.append((time.perf_counter() - t0) / steps * 1e3)
        self.barrier()
        dt = self.max_time(time.perf_counter() - t0)
        per = [[e0.elapsed_time(e1) for e0, e1 in dev_ev] for dev_ev in ev]
        kms = sum(sum(p) for p in per) / (steps * len(self.devs))
        info = {"per_shard_ms_per_step": self.gather_floats(own),
                "per_shard_kernel_ms": self.gather_floats([sum(p) / steps for p in per]),
                "kernel_ms_per_launch": per[0] if self.rank == 0 else None}
        return dt, kms, info

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


class Batch:
    """per-GPU resident buffers of one workload: symbols in, packed words / flags / sweep counts out"""

    def __init__(self, rig, H, cws, F):
        import numpy as np
        torch = rig.torch
        self.rig, self.H, self.cws, self.F = rig, H, cws, F
        self.n, self.nw = H.n, (H.n + 31) // 32
        self.y, self.bits, self.ok, self.its, self.cwp = {}, {}, {}, {}, {}
        pad = np.zeros((cws.shape[0], self.nw * 32), dtype=np.uint8)
        pad[:, :H.n] = cws
        packed = np.packbits(pad, axis=1, bitorder="little").view(np.int32).copy()
        for d in rig.devs:
            dev = "cuda:%d" % d.idx
            self.y[d.shard] = torch.empty((F, self.n), dtype=torch.float32, device=dev)
            self.bits[d.shard] = torch.zeros((F, self.nw), dtype=torch.int32, device=dev)
            self.ok[d.shard] = torch.zeros(F, dtype=torch.uint8, device=dev)
            self.its[d.shard] = torch.zeros(F, dtype=torch.int32, device=dev)
            self.cwp[d.shard] = torch.from_numpy(packed).to(dev)

    def gen_noise(self, decs, snr, seed=1):
        """fills y of every shard with the global frames [shard*F, (shard+1)*F) of the Philox AWGN stream"""
        import ctypes as C
        from acg_alp_ldpc_amd._lib import McCfg, check, lib
        for d in self.rig.devs:
            h, _ = decs[d.shard].handle(self.H)
            cfg = McCfg()
            cfg.frames, cfg.first_frame, cfg.snr, cfg.seed, cfg.noise = self.F, d.shard * self.F, snr, seed, 0
            cfg.codewords, cfg.n_codewords = self.cws.ctypes.data, self.cws.shape[0]
            check(lib().acg_ldpc_awgn_dev(h, C.byref(cfg), self.y[d.shard].data_ptr(), d.stream.cuda_stream))
        self.rig.sync_local()

    def step_fn(self, decs, snr, frames=None):
        F = self.F if frames is None else frames

        def step(d):
            s = d.shard
            decs[s].decode_batch_dev(self.H, self.y[s].data_ptr(), False, F, snr, self.bits[s].data_ptr(),
                                     self.ok[s].data_ptr(), self.its[s].data_ptr(), d.stream.cuda_stream)
        return step

    def quality(self, frames=None):
        """FER / mean sweeps of the batch currently in (bits, ok, its), all shards.  (QP-ADMM: ok is always 1; a wrong
        word is counted by the compare.)"""
        torch = self.rig.torch
        F = self.F if frames is None else frames
        tot = [0, 0, 0, 0]
        for d in self.rig.devs:
            s = d.shard
            with torch.cuda.device(d.idx):
                idx = (torch.arange(F, device=self.y[s].device, dtype=torch.int64) + s * self.F) % self.cwp[s].shape[0]
                good = (self.bits[s][:F] == self.cwp[s][idx]).all(dim=1) & (self.ok[s][:F] == 1)
                v = torch.stack([good.sum(), self.ok[s][:F].sum(), self.its[s][:F].sum()]).to(torch.int64).tolist()
            tot = [tot[0] + v[0], tot[1] + v[1], tot[2] + v[2], tot[3] + F]
        c, k, i, t = self.rig.sum_ints(tot)
        return {"fer": (t - c) / t, "undetected": k - c, "mean_iters": i / t, "frames": t}


def make_decoders(rig, ctor):
    return {d.shard: ctor(d.idx) for d in rig.devs}


def close_decoders(decs):
    for x in decs.values():
        x.close()


def decode_leg(rig, batch, ctor, snr, steps, warmup, frames=None, noise_seed=1):
    """time `steps` steps of one decoder configuration on every shard -> dict"""
    decs = make_decoders(rig, ctor)
    batch.gen_noise(decs, snr, noise_seed)
    F = batch.F if frames is None else frames
    dt, kms, info = rig.timed(batch.step_fn(decs, snr, F), steps, warmup)
    q = batch.quality(F)
    d0 = next(iter(decs.values()))
    lay = d0.layout(batch.H)
    desc = d0.describe(batch.H)
    close_decoders(decs)
    return {"value": rig.nshards * F * steps / dt, "unit": "frames/s", "ms_per_step": dt / steps * 1e3, "kernel_ms": kms,
            "steps": steps, "frames_per_gpu": F, "snr_db": snr, "fer": q["fer"], "undetected_errors": q["undetected"],
            "mean_iters": q["mean_iters"], "layout": lay, "instance": desc,
            "per_shard_ms_per_step": info["per_shard_ms_per_step"], "per_shard_kernel_ms": info["per_shard_kernel_ms"],
            "kernel_ms_per_launch": info["kernel_ms_per_launch"]}


def mc_leg(rig, H, cws, ctor, snr, F, steps, noise="device"):
    """acg_ldpc_mc_run (AWGN in the kernel, classification, D2H of the seven counters): one host thread per GPU, as the
    reference's multithread_experiment drives its workers (experiment.h:125-139)"""
    import acg_alp_ldpc_amd as A
    decs = make_decoders(rig, ctor)
    res, own = {}, {}

    def work(d, n, t0):
        for k in range(n):
            r = A.run_experiment(decs[d.shard], cws, H, snr, frames=F, first_frame=d.shard * F, noise=noise, seed=1 + k)
            res[d.shard] = r
        own[d.shard] = (time.perf_counter() - t0) / max(n, 1) * 1e3

    def run(n):
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(d, n, t0)) for d in rig.devs]
        [t.start() for t in th]
        [t.join() for t in th]

    run(1)
    rig.barrier()
    t0 = time.perf_counter()
    run(steps)
    rig.barrier()
    dt = rig.max_time(time.perf_counter() - t0)
    v = [0] * 7
    for r in res.values():
        v = [x + int(y) for x, y in zip(v, r.as_vector())]
    v = rig.sum_ints(v)
    close_decoders(decs)
    kms = sum(r.kernel_ms for r in res.values()) / max(1, len(res))
    return {"value": rig.nshards * F * steps / dt, "unit": "frames/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "kernel_ms": kms,
            "frames_per_gpu": F, "snr_db": snr, "fer": (v[2] - v[0]) / v[2], "pseudo_codewords": v[1],
            "mean_iters": v[6] / v[2], "raw_channel_errors_per_frame": v[3] / v[2], "counters": v,
            "per_shard_ms_per_step": rig.gather_floats([own[d.shard] for d in rig.devs])}


# ----------------------------------------------------------------------------------------------------------------
# workloads shared by the main run and the PMC probe child
def ctor_table(A, a):
    eng_s = A.ENGINE_STREAMED
    return {
        "bp_fused": lambda dev: A.BeliefPropagationDecoder(a.iters, early_exit=False, device=dev, lanes_per_frame=a.lanes),
        "bp_exit": lambda dev: A.BeliefPropagationDecoder(a.iters, early_exit=True, device=dev, lanes_per_frame=a.lanes),
        "bp_streamed": lambda dev: A.BeliefPropagationDecoder(a.iters, early_exit=False, device=dev, engine=eng_s),
        "ms_fused": lambda dev: A.MinSumDecoder(a.iters, 0.75, early_exit=False, device=dev, lanes_per_frame=a.lanes),
        "ms_exit": lambda dev: A.MinSumDecoder(a.iters, 0.75, early_exit=True, device=dev, lanes_per_frame=a.lanes),
        "ms_streamed": lambda dev: A.MinSumDecoder(a.iters, 0.75, early_exit=False, device=dev, engine=eng_s),
        # layered schedule: half the iterations for the same FER — a different algorithm, FER-level parity only
        "ms_layered": lambda dev: A.MinSumDecoder(a.iters // 2, 0.75, early_exit=False, device=dev, schedule=A.SCHEDULE_LAYERED),
        "ms_layered_exit": lambda dev: A.MinSumDecoder(a.iters // 2, 0.75, early_exit=True, device=dev, schedule=A.SCHEDULE_LAYERED),
        # the reference's sum-product check rule in the layered order: its FER at half its iterations (FER-level parity only)
        "bp_layered": lambda dev: A.BeliefPropagationDecoder(a.iters // 2, early_exit=False, device=dev, schedule=A.SCHEDULE_LAYERED),
        "bp_layered_exit": lambda dev: A.BeliefPropagationDecoder(a.iters // 2, early_exit=True, device=dev, schedule=A.SCHEDULE_LAYERED),
        "bp_layered_f16_exit": lambda dev: A.BeliefPropagationDecoder(a.iters // 2, early_exit=True, device=dev, schedule=A.SCHEDULE_LAYERED, precision=A.PREC_F16),
        "ms_layered_f16": lambda dev: A.MinSumDecoder(a.iters // 2, 0.75, early_exit=False, device=dev, schedule=A.SCHEDULE_LAYERED, precision=A.PREC_F16),
        "ms_layered_f16_exit": lambda dev: A.MinSumDecoder(a.iters // 2, 0.75, early_exit=True, device=dev, schedule=A.SCHEDULE_LAYERED, precision=A.PREC_F16),
        "qpadmm": lambda dev: A.QPADMMDecoder(a.alpha, a.mu, 100, 0.0, device=dev),          # eps 0: every frame runs 100 sweeps
        "qpadmm_exit": lambda dev: A.QPADMMDecoder(a.alpha, a.mu, 100, 1e-5, device=dev),
        "c5_block_ms": lambda dev: A.MinSumDecoder(50, 0.75, early_exit=False, device=dev),
        "c5_block_spa": lambda dev: A.BeliefPropagationDecoder(50, early_exit=False, device=dev),
        "c5_pair_f16_ms": lambda dev: A.MinSumDecoder(50, 0.75, early_exit=False, device=dev, precision=A.PREC_F16),
        "c5_streamed_ms": lambda dev: A.MinSumDecoder(50, 0.75, early_exit=False, device=dev, engine=eng_s),
    }


def load_h05(A, a):
    H = A.read_pcm(a.matrix)
    G, ok = H.get_orthogonal()
    assert ok
    return H, A.gen_random_codewords(G, 8192, 239239239)


def load_c5(A):
    import numpy as np
    H = A.ParityCheckMatrix(A.regular_ldpc(5000, 10000, 3, 6, seed=1))
    return H, np.zeros((1, 10000), dtype=np.uint8)   # all-zero codeword (both decoders are symmetric)


def pmc_probe_child(a):
    """run under `rocprofv3 --pmc ... -- python3 bench.py --pmc-probe`: every probed kernel exactly twice, in PROBE_ITEMS order"""
    import acg_alp_ldpc_amd as A
    rig = Rig(a)
    T = ctor_table(A, a)
    H, cws = load_h05(A, a)
    b = Batch(rig, H, cws, a.frames)
    items = [x for x in PROBE_ITEMS if not a.probe_items or x in a.probe_items.split(",")]
    for name in items:
        if name.startswith("c5_"):
            continue
        if name == "bp_mc":   # the Monte-Carlo kernel (AWGN + decode with the reference's stopping rule + classification)
            decs = make_decoders(rig, T["bp_exit"])
            for k in range(2):
                A.run_experiment(decs[0], cws, H, a.snr, frames=a.frames, noise="device", seed=1 + k)
            close_decoders(decs)
            continue
        decs = make_decoders(rig, T[name])
        b.gen_noise(decs, a.snr)
        for _ in range(2):
            b.step_fn(decs, a.snr)(rig.devs[0])
        rig.sync_local()
        close_decoders(decs)
    if any(x.startswith("c5_") for x in items):
        del b
        H5, cw5 = load_c5(A)
        b5 = Batch(rig, H5, cw5, a.c5_frames)
        for name in items:
            if not name.startswith("c5_"):
                continue
            decs = make_decoders(rig, T[name])
            b5.gen_noise(decs, 2.0)
            for _ in range(2):
                b5.step_fn(decs, 2.0)(rig.devs[0])
            rig.sync_local()
            close_decoders(decs)
    print("PMC_PROBE_DONE " + ",".join(items))


PROBE_KERNEL = {  # item -> (substring of the rocprofv3 kernel name, position among the probe's uses of that kernel)
    "bp_fused": ("bp_fused_kernel<float, 8, 32, 0, false, true, 12, false>", 0), "bp_exit": ("bp_fused_kernel<float, 8, 32, 0, false, true, 12, false>", 1),
    "bp_mc": ("bp_fused_kernel<float, 8, 32, 0, true, true, 12, false>", 0), "bp_streamed": ("bp_streamed_ring_kernel<0, false, false>", 0),
    "ms_streamed": ("bp_streamed_ring_kernel<1, false, false>", 0), "ms_layered": ("bp_layered_kernel<20, 2, false, float, 1, false>", 0), "ms_layered_f16": (("bp_layered_kernel<20, 4, false, _Float16, 1, false>", "bp_layered_kernelILi20ELi4ELb0EDF16_Li1ELb0EE"), 0),   # (rocprofv3 leaves _Float16 mangled)
    "bp_layered": ("bp_layered_kernel<20, 2, false, float, 0, false>", 0),
    "qpadmm": ("admm_block_kernel<double, false, 3, true>", 0),
    "c5_block_ms": ("bp_block_kernel<float, 1024, 1, false, false, true, false, true>", 0), "c5_pair_f16_ms": ("bp_pair_kernel<1024, true>", 0), "c5_streamed_ms": ("bp_streamed_ring_kernel<1, true, false>", -1),   # (-1: the last two dispatches — the workspace probes launch this kernel too)
}


def pmc_one_pass(a, tag, ctrs, left, env, out):
    """one rocprofv3 --pmc run of the probe child; fills out["items"]; -> error text or None"""
    import csv
    import glob
    import signal
    tmp = tempfile.mkdtemp(prefix="acg_pmc_%s_" % tag, dir="/tmp")
    cmd = [os.environ.get("ACG_BENCH_ROCPROF", "rocprofv3"), "--kernel-trace", "--pmc"] + ctrs + ["--output-format", "csv", "-d", tmp, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-probe", "--frames", str(a.frames), "--c5-frames",
           str(a.c5_frames), "--snr", str(a.snr), "--iters", str(a.iters), "--alpha", str(a.alpha), "--mu", str(a.mu),
           "--matrix", a.matrix, "--lanes", str(a.lanes)]
    try:
        p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                             start_new_session=True)
        try:
            log, _ = p.communicate(timeout=min(left, 240))
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)   # exactly the process group this function started
            p.communicate()
            return "counter pass '%s' timed out" % tag
        if p.returncode != 0 or "PMC_PROBE_DONE" not in log:
            return "counter pass '%s' failed (rc %s): %s" % (tag, p.returncode, log[-300:])
        files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            return "counter pass '%s' wrote no counter_collection.csv" % tag
        rows = []
        for f in files:
            rows += list(csv.DictReader(open(f)))
        per = {}   # kernel name -> dispatch id -> counter -> value (summed over the rows of one dispatch)
        for r in rows:
            did = int(r.get("Dispatch_Id", r.get("Dispatch_ID", 0)))
            dct = per.setdefault(r["Kernel_Name"], {}).setdefault(did, {})
            dct[r["Counter_Name"]] = dct.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        dur = {}   # kernel name -> dispatch id -> ms (kernel-trace of the same pass)
        for f in glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                try:
                    dur.setdefault(r["Kernel_Name"], {})[int(r.get("Dispatch_Id", 0))] = \
                        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                except (KeyError, ValueError):
                    pass
        for item, (pat, pos) in PROBE_KERNEL.items():
            ks = [k for k in per if any(q in k for q in ((pat,) if isinstance(pat, str) else pat))]
            if not ks:
                continue
            disp = sorted(per[ks[0]].items())
            mine = disp[-2:] if pos < 0 else disp[2 * pos:2 * pos + 2]
            for c in ctrs:
                vals = [d[c] for _, d in mine if c in d]
                if vals:
                    out["items"][item][c] = sum(vals) / len(vals)
                    out["items"][item][c + "_per_dispatch"] = vals          # both launches, not only their mean
            ms = [dur.get(ks[0], {}).get(did) for did, _ in mine]
            if all(x is not None for x in ms) and ms:
                out["items"][item]["DISPATCH_MS_pass_" + tag] = ms
    except FileNotFoundError:
        return "rocprofv3 not found"
    finally:
        subprocess.call(["rm", "-rf", tmp])
    return None


def pmc_collect(a, budget_s):
    """Three rocprofv3 counter passes (each counter group alone, kernel-trace only) over the probe child.
    -> {"items": {item: {counter: mean per launch}}, ...} or {"error": ...}"""
    out = {"source": "rocprofv3 --pmc passes run by this bench.py process (child: bench.py --pmc-probe)", "csrc_sha": csrc_sha(),
           "frames": a.frames, "c5_frames": a.c5_frames, "items": {k: {} for k in PROBE_ITEMS}}
    t_start = time.time()
    env = dict(os.environ)
    env["TMPDIR"] = "/tmp"
    for tag, ctrs in PMC_PASSES:
        left = budget_s - (time.time() - t_start)
        err = None
        if left < 30:
            err = "time budget for the counter passes exhausted before pass '%s'" % tag
        else:
            err = pmc_one_pass(a, tag, ctrs, left, env, out)
        if err and tag in PMC_OPTIONAL:
            out.setdefault("notes", []).append(err)
        elif err:
            out["error"] = err
            break
    # static instruction mix of the probed kernels, priced by operation class (tools/valu_mix.py): carried with the counters
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import valu_mix
        mix = valu_mix.static_mix({item: (pat if isinstance(pat, str) else pat[0]) for item, (pat, _) in PROBE_KERNEL.items()})
        for item, m in mix.items():
            out["items"][item].update({"STATIC_VALU_FULL_RATE": m["full_rate"], "STATIC_VALU_HALF_RATE": m["half_rate"],
                                       "STATIC_VALU_TRANS": m["transcendental"],
                                       "STATIC_CYC_PER_NON_TRANS": m["cycles_per_non_transcendental"]})
    except Exception as e:  # the lower bound stands alone then
        out.setdefault("notes", []).append("static instruction mix unavailable: %r" % (e,))
    out["seconds"] = time.time() - t_start
    return out


def pmc_lookup(pmc, item):
    if pmc and pmc.get("items", {}).get(item):
        return pmc["items"][item], pmc.get("source")
    return None, None


def roofline_fused(c, src, kms, F, bpf, fp64=False):
    """LDS-resident kernels: VALU-issue and LDS-array utilisation from the counters of this launch shape"""
    r = {"streamed_equiv_frac": F * bpf / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "streamed_equiv_bytes_per_frame": bpf,
         "kernel_ms": kms, "counters_from": src}
    if not c or "SQ_INSTS_VALU" not in c:
        r.update({"bound": "valu_issue", "achieved": None, "peak": N_SIMD * NOMINAL_CLK_HZ / 1e9, "unit": "Gcycle/s", "frac": None,
                  "traffic": None, "note": "no counter pass available for this kernel (rocprofv3 failed and no committed "
                                           "pass matches these sources): utilisation not stated"})
        return r
    nv, nt = c["SQ_INSTS_VALU"], c.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
    sec = kms * 1e-3
    # GRBM_GUI_ACTIVE counts per XCD (8 of them): cycles the chip was busy in the profiled launch ~ clock x duration
    n64 = None
    if fp64 and "SQ_INSTS_VALU_FMA_F64" in c:   # fp64 add / mul / fma counted by class; the rest (moves, integer, compares, max) at 2
        n64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"] + c.get("SQ_INSTS_VALU_TRANS_F64", 0.0)
        valu_cyc = (nv - nt - n64) * CYC_VALU + n64 * CYC_VALU_F64 + nt * CYC_TRANS
    else:
        valu_cyc = (nv - nt) * CYC_VALU + nt * CYC_TRANS   # no class split available: everything at 2 (still guaranteed)
    valu = valu_cyc / sec / 1e9                     # Gcycle/s of VALU issue actually consumed, whole chip
    lds = c.get("SQ_LDS_IDX_ACTIVE", 0.0) / sec / 1e9
    peak_valu, peak_lds = N_SIMD * NOMINAL_CLK_HZ / 1e9, N_CU * NOMINAL_CLK_HZ / 1e9
    fv, fl = valu / peak_valu, lds / peak_lds
    # The same counts priced with the kernel's own static mix of full-rate (2 cycles) and half-rate instructions (4 cycles:
    # every fp64 operation, compares, min/max, selects on SGPR masks, left shifts, shift-adds, and-ors, SDWA/DPP, packed f16,
    # ...; tools/valu_mix.py, per-operation costs from tools/microbench/valu_op_rates.hip): `frac_by_op_class`, an estimate.
    f_mix = None
    if c.get("STATIC_CYC_PER_NON_TRANS"):
        f_mix = ((nv - nt) * c["STATIC_CYC_PER_NON_TRANS"] + nt * CYC_TRANS) / sec / 1e9 / peak_valu
    f_upper = ((nv - nt) * 4.0 + nt * CYC_TRANS) / sec / 1e9 / peak_valu
    # frac = the figure the counters and the guide's cycle constants GUARANTEE (2 cycles per VALU wave-instruction, 8 per
    # transcendental; fp64 arithmetic by class at 4).  The operation-class estimate is reported beside it, never as frac.
    traffic = None
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:   # rocprofv3 reports KiB; gfx950: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM)
        traffic = c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024
    if fv >= fl:
        r.update({"bound": "valu_issue", "achieved": valu, "peak": peak_valu, "unit": "Gcycle/s", "frac": fv})
    else:
        r.update({"bound": "lds", "achieved": lds, "peak": peak_lds, "unit": "Gcycle/s", "frac": fl})
    r.update({"traffic": traffic, "valu_issue_frac": fv, "lds_array_frac": fl,
              "lds_bank_conflict_share": (c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else None,
              "wave_wait_share": (c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else None,
              "valu_insts_per_launch": nv, "trans_insts_per_launch": nt, "f64_arith_insts_per_launch": n64,
              "frac_by_op_class": f_mix, "valu_issue_upper_frac": min(f_upper, 1.0),
              "frac_basis": "measured wave-instruction counts priced with the guide's cycle constants (2 per VALU instruction, 8 per "
                            "transcendental%s): guaranteed lower bound of the VALU-issue utilisation" % (", 4 per fp64 add/mul/fma" if n64 is not None else ""),
              "frac_by_op_class_basis": ("the same counts with the non-transcendental instructions priced by the kernel's STATIC mix of "
                                         "full-rate (2) and half-rate (4) operations (%d / %d / %d transcendental instructions in the kernel "
                                         "text, tools/valu_mix.py, prices from profiles/r02_valu_op_rates.txt): an estimate, the executed mix "
                                         "is not the static mix" % (c["STATIC_VALU_FULL_RATE"], c["STATIC_VALU_HALF_RATE"], c["STATIC_VALU_TRANS"])
                                         if f_mix is not None else None),
              "model": "VALU: (%s x other + %g x transcendental wave-instructions) / (1024 SIMDs x 2.4 GHz x kernel time)%s; "
                       "LDS: SQ_LDS_IDX_ACTIVE / (256 CUs x 2.4 GHz x kernel time); nominal clock, so both are lower bounds of the "
                       "utilisation at the clock actually sustained" % ("%g" % CYC_VALU, CYC_TRANS,
                                                                       (" — fp64 add/mul/fma (SQ_INSTS_VALU_*_F64) at 4 cycles, everything else at 2" if n64 is not None else
                                                                        " — fp64 class counters unavailable: every VALU op at 2") if fp64 else "")
})
    return r


def roofline_hbm(c, src, kms, F, bpf, working_set=None):
    achieved = F * bpf / (kms * 1e-3) / 1e9
    traffic = None
    if c and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        traffic = c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024
    resident = bool(working_set) and working_set <= (256 << 20)
    # Slabs that fit in the 256 MiB Infinity Cache are served on-die: what FETCH_SIZE / WRITE_SIZE count there is L2 <-> fabric
    # traffic, NOT HBM traffic, so the leg is labelled "fabric_l2"; the 8 TB/s HBM figure stays as the yardstick (`peak`)
    # because the guide states no separate peak for the Infinity Cache path.
    return {"bound": "fabric_l2" if resident else "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_over_algorithmic": (traffic / (F * bpf)) if traffic else None, "kernel_ms": kms,
            "bytes_per_frame": bpf, "counters_from": src, "slab_working_set_bytes": working_set,
            "infinity_cache_resident": resident if working_set else None,
            "note": ("the message slabs of all resident workgroups (%.0f MB) fit in the 256 MiB Infinity Cache: the traffic counted here is "
                     "L2 <-> fabric traffic, most of which never reaches the HBM stacks; frac = achieved / 8 TB/s is a yardstick, not HBM "
                     "utilisation" % (working_set / 1e6)) if resident
                    else ("slabs (%.0f MB) exceed the Infinity Cache: HBM traffic.  Measured ceiling of the bare access pattern (read a 256-byte "
                          "line, write it back) on HBM-resident slabs: 5.0-5.5 TB/s (profiles/r02_slab_stream.txt)" % (working_set / 1e6)) if working_set else None,
            "model": "SURVEY 8(d) streamed model: %d B/frame x %d frames / %.3f ms mean kernel time; the messages live in memory, "
                     "so this IS memory-side traffic (traffic = FETCH_SIZE x2 + WRITE_SIZE of a rocprofv3 pass over the same launch)" % (bpf, F, kms)}


# ----------------------------------------------------------------------------------------------------------------
# The compact line: what the driver's record keeps (its stdout tail is 8 KB).  Pure function of the detail object.
COMPACT_LIMIT = 4096


def _r(x, sig=5):
    """float -> `sig` significant digits (ints and None pass)"""
    if x is None or isinstance(x, (bool, int, str)):
        return x
    try:
        return float("%.*g" % (sig, float(x)))
    except (TypeError, ValueError):
        return None


def _triple(leg, rl=None):
    """{value, frac, bound} of a side leg (frac / bound None when the leg has no counter-backed roofline)"""
    if not leg:
        return None
    rl = rl if rl is not None else leg.get("roofline")
    v = _r(leg.get("value"), 6)
    t = {"value": int(v) if isinstance(v, float) and v >= 1e5 else v, "frac": _r((rl or {}).get("frac"), 4)}
    if (rl or {}).get("bound") is not None:      # (legs without a counter pass: frac null, no bound)
        t["bound"] = rl["bound"]
    if leg.get("fer") is not None:
        t["fer"] = _r(leg["fer"], 4)
    return t


def _cpu(c):
    if not c:
        return None
    return {"value": _r(c.get("value"), 6), "unit": c.get("unit", "frames/s"), "cores": c.get("cores"), "kind": c.get("kind"),
            "frames": c.get("frames"), "cores_visible": c.get("cores_visible"), "busy_cores_mean": _r(c.get("busy_cores_mean"), 4),
            "per_core": _r(c.get("frames_per_s_per_core"), 4), "survey_1thread": c.get("survey_single_thread_frames_per_s"),
            "sample": "%s frames of the same H05/AWGN workload, %s 1-thread processes, own stopping rule" % (c.get("frames"), c.get("cores"))}


def compact_line(d):
    """detail object of a run -> the compact object of the final stdout line (json.dumps of it stays < COMPACT_LIMIT bytes)"""
    rl = d.get("roofline") or {}
    cfg = d.get("config") or {}
    out = {
        "metric": d.get("metric"), "value": _r(d.get("value"), 7), "unit": d.get("unit"), "n_gpus": d.get("n_gpus"),
        "steps": d.get("steps"), "warmup": d.get("warmup"), "ms_per_step": _r(d.get("ms_per_step"), 6),
        "higher_is_better": True, "scaling": d.get("scaling", "weak"), "vs_baseline": d.get("vs_baseline"),
        "dtype": d.get("dtype"), "data": d.get("data", "synthetic"),
        "config": {"workload": cfg.get("workload_short") or (cfg.get("workload") or "")[:160], "frames_per_gpu": cfg.get("frames_per_gpu"),
                   "iters": cfg.get("iters"), "early_exit": cfg.get("early_exit"), "snr_db": cfg.get("snr_db"),
                   "launcher": (cfg.get("launcher") or "")[:60], "engine": cfg.get("engine")},
        "fer": _r(d.get("fer"), 4), "mean_exit_iter": _r(d.get("mean_exit_iter"), 4),
        "roofline": {"bound": rl.get("bound"), "frac": _r(rl.get("frac"), 4), "achieved": _r(rl.get("achieved"), 5),
                     "peak": _r(rl.get("peak"), 5), "unit": rl.get("unit"), "traffic": _r(rl.get("traffic"), 5),
                     "kernel_ms": _r(rl.get("kernel_ms"), 5), "streamed_equiv_frac": _r(rl.get("streamed_equiv_frac"), 4),
                     "frac_by_op_class": _r(rl.get("frac_by_op_class", rl.get("valu_issue_frac_by_op_class")), 4), "lds_array_frac": _r(rl.get("lds_array_frac"), 4)},
    }
    if d.get("per_shard_ms_per_step"):
        out["per_shard_ms_per_step"] = [_r(x, 5) for x in d["per_shard_ms_per_step"]]
    if "cpu_baseline" in d:
        out["cpu_baseline"] = _cpu(d.get("cpu_baseline"))
    if d.get("cpu_baseline_qpadmm"):
        out["cpu_baseline_qpadmm"] = _cpu(d["cpu_baseline_qpadmm"])
    legs = {}
    ee = d.get("early_exit") or {}
    for k, v in ee.items():
        legs["early_exit_" + k] = _triple(v)
    mc = d.get("monte_carlo") or {}
    for k, v in mc.items():
        if isinstance(v, dict) and "value" in v:
            legs["mc_" + k] = _triple(v)       # acg_ldpc_mc_run: SURVEY 8(d)'s definitional metric (noise + decode + D2H)
    ms = d.get("minsum_0.75") or {}
    for k in ("fixed", "early_exit", "layered_fixed", "layered_exit", "layered_f16_fixed", "layered_f16_exit"):
        if ms.get(k):
            legs["minsum_" + k + " (parity unpinned)"] = _triple(ms[k])
    sl = d.get("sumproduct_layered") or {}
    for k in ("fixed", "early_exit", "f16_early_exit"):
        if sl.get(k):
            legs["spa_layered_" + k + " (FER-level parity)"] = _triple(sl[k])
    st = d.get("streamed") or {}
    for k in ("sum_product", "minsum_0.75"):
        if st.get(k):
            legs["streamed_h05_" + k + (" (parity unpinned)" if "minsum" in k else "")] = _triple(st[k])
    c2 = d.get("configs[2]") or {}
    for k, name in (("fixed_100_sweeps", "configs[2]_qpadmm_fixed"), ("residual_exit_1e-5", "configs[2]_qpadmm_exit")):
        if c2.get(k):
            legs[name] = _triple(c2[k])
    c4 = d.get("configs[4]") or {}
    for k, v in c4.items():
        if isinstance(v, dict) and "value" in v:
            legs["configs[4]_" + k + (" (parity unpinned)" if "minsum" in k else "")] = _triple(v)
    if legs:
        out["legs"] = legs
    if d.get("pmc_note"):
        out["pmc_note"] = d["pmc_note"][:120]
    out["detail"] = "bench_detail.json (cwd) and the 'BENCH_DETAIL ' stdout line"
    # the size is a contract: shed the least important fields rather than overflow the driver's tail
    for drop in ("pmc_note", "per_shard_ms_per_step"):
        if len(json.dumps(out)) < COMPACT_LIMIT:
            break
        if drop == "per_shard_ms_per_step" and out.get(drop) and len(out[drop]) <= 8:
            continue
        out.pop(drop, None)
    if len(json.dumps(out)) >= COMPACT_LIMIT and "legs" in out:
        for k in list(out["legs"]):
            out["legs"][k] = {"value": out["legs"][k]["value"], "frac": out["legs"][k]["frac"]} if out["legs"][k] else None
    while len(json.dumps(out)) >= COMPACT_LIMIT and out.get("legs"):
        out["legs"].popitem()
    return out


def emit(detail, path="bench_detail.json"):
    """full object -> ./bench_detail.json and an earlier stdout line; compact object -> the LAST stdout line"""
    try:
        with open(path, "w") as f:
            json.dump(detail, f)
    except OSError as e:
        sys.stderr.write("bench.py: could not write %s: %r\n" % (path, e))
    sys.stderr.flush()
    sys.stdout.write("BENCH_DETAIL " + json.dumps(detail) + "\n")
    sys.stdout.write(json.dumps(compact_line(detail)) + "\n")
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--inproc", type=int, default=0, help="drive N GPUs from ONE process (no torch.distributed)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=1 << 20, help="frames per GPU per step (configs[1], configs[2]: 1M)")
    ap.add_argument("--c5-frames", type=int, default=32768, help="frames per GPU per step of configs[4] (262144 over 8 GPUs)")
    ap.add_argument("--snr", type=float, default=-2.0)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--matrix", default=os.path.join(ROOT, "data", "H05.txt"))
    ap.add_argument("--lanes", type=int, default=0, help="lanes per frame of the fused BP kernels (0 = library default)")
    ap.add_argument("--alpha", type=float, default=1.95)
    ap.add_argument("--mu", type=float, default=0.5)
    ap.add_argument("--side-steps", type=int, default=3, help="timed steps of every side measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only")
    ap.add_argument("--all", action="store_true", help="N > 1: run every side leg too (default at N > 1: headline, early exit, Monte-Carlo)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (default: on at N=1)")
    ap.add_argument("--write-pmc", default=None, help="also store this run's counter passes here (profiles/pmc_rNN.json)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-time target of each CPU baseline sample")
    ap.add_argument("--detail-out", default="bench_detail.json", help="where the full object goes (default: cwd)")
    ap.add_argument("--pmc-probe", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--probe-items", default="", help=argparse.SUPPRESS)
    a = ap.parse_args()

    import acg_alp_ldpc_amd as A
    if not A.device_available():
        raise SystemExit("bench.py needs a HIP device; the product has no CPU path")
    if a.pmc_probe:
        return pmc_probe_child(a)

    rig = Rig(a)
    T = ctor_table(A, a)
    H, cws = load_h05(A, a)
    n, E, F = H.n, H.E, a.frames
    batch = Batch(rig, H, cws, F)
    ss = max(1, a.side_steps)
    single = rig.nshards == 1
    # N > 1 is the scaling curve of the METRIC: headline + early exit + Monte-Carlo legs.  The other legs (streamed engine,
    # QP-ADMM, configs[4] with its 5.3 GB of slabs per rank) are single-GPU characterisations and only run with --all.
    core_legs = not a.no_extras
    side_legs = core_legs and (single or a.all)

    # ---- headline: fixed 50 iterations, SNR a.snr -------------------------------------------------------
    head = decode_leg(rig, batch, T["bp_fused"], a.snr, a.steps, a.warmup)
    bpf = bp_bytes_per_frame(n, E, a.iters)
    out = {
        "metric": "decoded frames/sec (+ FER@SNR) for H05.txt 50-iter BP", "value": head["value"], "unit": "frames/s",
        "n_gpus": rig.nshards, "steps": a.steps, "warmup": a.warmup, "ms_per_step": head["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: %s (%dx%d, E=%d) sum-product BP, %d iterations FIXED (no early exit), %d AWGN "
                               "frames per GPU per step at Es/N0 %.1f dB, inputs/outputs resident in HBM"
                               % (os.path.basename(a.matrix), H.m, n, E, a.iters, F, a.snr),
                   "workload_short": "configs[1]: %s %dx%d E=%d sum-product BP, %d iterations fixed, %d AWGN frames/GPU/step, HBM-resident"
                                     % (os.path.basename(a.matrix), H.m, n, E, a.iters, F),
                   "engine": "fused (messages in LDS)", "frames_per_gpu": F, "snr_db": a.snr, "iters": a.iters,
                   "early_exit": False, "launcher": rig.launcher,
                   "sharding": "frames [s*F,(s+1)*F) per GPU, no collective on the data path; control plane: %s"
                               % ("gloo (CPU tensors)" if rig.dist and rig.backend != "nccl" else ("nccl" if rig.dist else "none")),
                   "layout": head["layout"], "instance": head["instance"]},
        "fer": head["fer"], "undetected_errors": head["undetected_errors"], "mean_exit_iter": head["mean_iters"],
        "per_shard_ms_per_step": head["per_shard_ms_per_step"], "per_shard_kernel_ms": head["per_shard_kernel_ms"],
        "kernel_ms_per_launch": head["kernel_ms_per_launch"],
    }

    # ---- counter passes of this run (N = 1 only) --------------------------------------------------------
    pmc = None
    if single and not a.no_pmc and not a.no_extras:
        rig.sync_local()
        pmc = pmc_collect(a, budget_s=300)
        if a.write_pmc and "error" not in pmc:
            json.dump(pmc, open(a.write_pmc, "w"), indent=1)
    if pmc is None or "error" in (pmc or {}):
        # fall back to the committed passes, but only if they were taken from exactly these kernel sources
        err = (pmc or {}).get("error")
        pmc = None
        for f in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
            if f.startswith("pmc_r") and f.endswith(".json"):
                try:
                    cand = json.load(open(os.path.join(ROOT, "profiles", f)))
                except Exception:
                    continue
                if cand.get("csrc_sha") == csrc_sha() and cand.get("frames") == F and cand.get("c5_frames") == a.c5_frames:
                    pmc = cand
                    pmc["source"] = "committed profiles/%s (same kernel sources: csrc_sha %s)" % (f, cand["csrc_sha"])
                    break
        out["pmc_note"] = ("live counter passes unavailable (%s); " % err if err else "") + \
                          ("using " + pmc["source"] if pmc else "no committed pass matches these kernel sources (stale passes are refused)")
    c, src = pmc_lookup(pmc, "bp_fused")
    out["roofline"] = roofline_fused(c, src, head["kernel_ms"], F, bpf)

    if core_legs:
        # ---- reference semantics (early exit) at a.snr and +2 dB ------------------------------------------
        ee = {}
        for snr in (a.snr, 2.0):
            r = decode_leg(rig, batch, T["bp_exit"], snr, ss, 1)
            if snr == a.snr:
                c, src = pmc_lookup(pmc, "bp_exit")
                r["roofline"] = roofline_fused(c, src, r["kernel_ms"], F, bp_bytes_per_frame(n, E, r["mean_iters"]))
            else:
                r["streamed_equiv_frac"] = F * bp_bytes_per_frame(n, E, r["mean_iters"]) / (r["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            ee["%+.1fdB" % snr] = r
        out["early_exit"] = ee
        # ---- Monte-Carlo mode: the metric as SURVEY §8(d) defines it -------------------------------------
        mc_main = mc_leg(rig, H, cws, T["bp_exit"], a.snr, F, ss)
        c, src = pmc_lookup(pmc, "bp_mc")
        mc_main["roofline"] = roofline_fused(c, src, mc_main["kernel_ms"], F, bp_bytes_per_frame(n, E, mc_main["mean_iters"]))
        out["monte_carlo"] = {
            "definition": "acg_ldpc_mc_run: AWGN generated on the device, decode with the reference's stopping rule, "
                          "classification against the sent word, D2H of the seven counters; wall time of the calls",
            "bp50_%+.1fdB" % a.snr: mc_main,
            "bp50_+2.0dB": mc_leg(rig, H, cws, T["bp_exit"], 2.0, F, ss),
        }
    if side_legs:
        out["monte_carlo"]["qpadmm100_%+.1fdB" % a.snr] = mc_leg(rig, H, cws, T["qpadmm_exit"], a.snr, F // 4, ss)
        # ---- north_star's named variant: normalised min-sum (NOT in the reference: parity unpinned) --
        out["minsum_0.75"] = {"fixed": decode_leg(rig, batch, T["ms_fused"], a.snr, ss, 1),
                              "early_exit": decode_leg(rig, batch, T["ms_exit"], a.snr, ss, 1),
                              "note": "min-sum is not in the reference: parity unpinned"}
        for key, name in (("ms_layered", "layered_fixed"), ("ms_layered_exit", "layered_exit"), ("ms_layered_f16", "layered_f16_fixed"),
                          ("ms_layered_f16_exit", "layered_f16_exit")):
            # layered schedule: half the sweeps for the same FER, FER-level parity only; f16 = messages stored in half precision
            r = decode_leg(rig, batch, T[key], a.snr, ss, 1)
            if key in ("ms_layered", "ms_layered_f16"):
                c, src = pmc_lookup(pmc, key)
                r["roofline"] = roofline_fused(c, src, r["kernel_ms"], F, bp_bytes_per_frame(n, E, a.iters // 2, b=4 if key == "ms_layered" else 2))
            out["minsum_0.75"][name] = r
        # ---- sum-product with the layered schedule: the reference's check rule (bp.h:49-57), another message order ----
        sl = {"note": "layered schedule: a different algorithm from BeliefPropagationDecoder (bp.h:183-199 floods) — FER-level parity only: "
                      "FER <= the flooding decoder's at half the iterations (tests/test_layered.py)"}
        for key, name in (("bp_layered", "fixed"), ("bp_layered_exit", "early_exit"), ("bp_layered_f16_exit", "f16_early_exit")):
            r = decode_leg(rig, batch, T[key], a.snr, ss, 1)
            if key == "bp_layered":
                c, src = pmc_lookup(pmc, key)
                r["roofline"] = roofline_fused(c, src, r["kernel_ms"], F, bp_bytes_per_frame(n, E, a.iters // 2))
            sl[name] = r
        out["sumproduct_layered"] = sl
        # the same through the Monte-Carlo loop (AWGN kernel -> decode -> classification kernel, all on the device)
        out["monte_carlo"]["bp%d_layered_%+.1fdB" % (a.iters // 2, a.snr)] = mc_leg(rig, H, cws, T["bp_layered_exit"], a.snr, F, ss)
        # ---- the HBM-resident engine: messages [edge][frame] in HBM, one lane per frame ------------------
        st = {}
        for key, item in (("sum_product", "bp_streamed"), ("minsum_0.75", "ms_streamed")):
            r = decode_leg(rig, batch, T[item], a.snr, ss, 1)
            c, src = pmc_lookup(pmc, item)
            r["roofline"] = roofline_hbm(c, src, r["kernel_ms"], F, bpf, working_set=r["layout"]["grid_blocks"] * ((E + n) * 256 + n * 16))
            st[key] = r
        st["workload"] = "configs[1] on the streamed engine: same frames, 50 iterations fixed, messages in memory"
        out["streamed"] = st
        # ---- configs[2]: QP-ADMM(alpha, mu) fp64, 100 sweeps, 1M frames ----------------------------------
        sh = H.admm_shape()
        bpf_q = admm_bytes_per_frame(n, sh["n_con"], sh["n_var"], 100)
        q_fixed = decode_leg(rig, batch, T["qpadmm"], a.snr, ss, 1)
        c, src = pmc_lookup(pmc, "qpadmm")
        q_fixed["roofline"] = roofline_fused(c, src, q_fixed["kernel_ms"], F, bpf_q, fp64=True)
        q_exit = decode_leg(rig, batch, T["qpadmm_exit"], a.snr, ss, 1)
        out["configs[2]"] = {"workload": "H05 QP-ADMM(%g, %g) fp64, 100 sweeps, %d frames per GPU per step at %.1f dB"
                                         % (a.alpha, a.mu, F, a.snr), "dtype": "f64",
                             "fixed_100_sweeps": q_fixed, "residual_exit_1e-5": q_exit}
        # ---- configs[4]: synthetic (3,6)-regular 5000 x 10000, min-sum 50 iterations ---------------------
        del batch
        rig.torch.cuda.empty_cache()
        H5, cw5 = load_c5(A)
        b5 = Batch(rig, H5, cw5, a.c5_frames)
        bpf5 = bp_bytes_per_frame(H5.n, H5.E, 50)
        c5 = {"workload": "configs[4]: synthetic (3,6)-regular %dx%d (E=%d), 50 iterations FIXED, %d frames per GPU per step "
                          "at +2.0 dB, all-zero codeword" % (H5.m, H5.n, H5.E, a.c5_frames)}
        for key, item, hbm in (("fused_block_minsum", "c5_block_ms", False), ("fused_pair_f16_minsum", "c5_pair_f16_ms", False),
                               ("fused_block_sum_product", "c5_block_spa", False), ("streamed_minsum", "c5_streamed_ms", True)):
            r = decode_leg(rig, b5, T[item], 2.0, ss + (3 if hbm else 0), 1)
            c, src = pmc_lookup(pmc, item)
            if hbm:
                r["roofline"] = roofline_hbm(c, src, r["kernel_ms"], a.c5_frames, bpf5,
                                             working_set=min(r["layout"]["grid_blocks"], (a.c5_frames + 63) // 64) * ((H5.E + H5.n) * 256 + H5.n * 16))
                if c:   # the same kernel under the counter passes: per-dispatch duration of every pass next to the timed launches
                    r["counter_pass_dispatch_ms"] = {k[len("DISPATCH_MS_pass_"):]: v for k, v in c.items() if k.startswith("DISPATCH_MS_pass_")}
            elif c or item in ("c5_block_ms", "c5_pair_f16_ms"):
                r["roofline"] = roofline_fused(c, src, r["kernel_ms"], a.c5_frames, bpf5)
            c5[key] = r
        out["configs[4]"] = c5
        del b5
    elif core_legs and not single:
        out["legs_note"] = "N > 1: headline, early-exit and Monte-Carlo legs only (the metric's scaling curve); --all runs the single-GPU characterisation legs too"

    # ---- CPU baselines (rank 0, N = 1 only) -------------------------------------------------------------
    if single and not a.no_cpu_baseline:
        import multiprocessing as mp
        Hd = H.dense()
        cores, core_info = usable_cores(mp.get_context("spawn"))
        out["cpu_baseline"] = cpu_baseline("bp", Hd, cws, a.snr, a.iters, a.cpu_seconds, cores=cores, core_info=core_info)
        out["cpu_baseline_qpadmm"] = cpu_baseline("qpadmm", Hd, cws, a.snr, 100, a.cpu_seconds, a.alpha, a.mu, cores=cores, core_info=core_info)
    elif single:
        out["cpu_baseline"] = None
    if pmc:
        out["pmc"] = {"source": pmc.get("source"), "csrc_sha": pmc.get("csrc_sha"), "seconds": pmc.get("seconds"),
                      "passes": [t for t, _ in PMC_PASSES], "items": pmc["items"]}

    if rig.rank == 0:
        emit(out, a.detail_out)
    rig.close()


if __name__ == "__main__":
    main()
