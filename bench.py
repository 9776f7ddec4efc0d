#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its config: decoded frames/s for data/H05.txt, 50-iteration
sum-product BP, 1M synthetic AWGN frames per GPU (BASELINE configs[1]), 1/2/4/8 MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (acg_ldpc_decode_batch_dev) over one resident batch of
`--frames` noisy frames per GPU.  The batch (channel symbols, fp32) is generated on the device
before the timed region; inputs and outputs stay in HBM.  Frames are independent, so rank r owns
global frames [r*F, (r+1)*F) — weak scaling, no data-path collective; the only
cross-rank traffic is the barrier / MAX of the timing and a sum of a few counters.

Headline `value`: FIXED WORK — every frame runs all 50 flooding iterations (output latched at its
first zero syndrome), i.e. nothing is skipped; this is the figure the streamed-message roofline
model of SURVEY §8(d) (745,155 B/frame) is written for.  The reference's own semantics (stop a
frame at its first zero syndrome, bp.h:195-196) is timed too and reported under "early_exit",
next to the reference CPU decoder ("cpu_baseline"), which only exists in that form.

The roofline object is the streamed-equivalent HBM figure: the fused kernel keeps messages in
LDS, so `achieved` may exceed the HBM peak and is NOT HBM utilisation;
`traffic` is the PMC-measured HBM traffic per launch when profiles/ holds one for this config.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def bp_bytes_per_frame(n, E, iters, b=4, b_in=4):
    """B_bp = n*b_in + I*(4E + n)*b + ceil(n/8)"""
    return n * b_in + iters * (4 * E + n) * b + (n + 7) // 8


def admm_bytes_per_frame(n, n_con, n_var, iters, b=8, b_in=4):
    """B_admm = n*b_in + I*(5C + 3V)*b + ceil(n/8)"""
    return n * b_in + iters * (5 * n_con + 3 * n_var) * b + (n + 7) // 8


def cpu_baseline_worker(args):
    """one process = one single-threaded reference decoder (BP's global node counter forbids threads)"""
    kind, Hm, y, snr, max_iter = args
    if kind == "reference":
        from oracle.pyoracle import Ref
        d = Ref()
    else:
        from oracle.pyoracle import Oracle
        d = Oracle()
    t0 = time.time()
    bits, ok, _ = d.bp_decode(Hm, y, snr, max_iter)
    return time.time() - t0, int(ok.sum())


def cpu_baseline(Hm, cws, snr, max_iter, per_proc):
    """reference (oracle/_ref, built from /root/reference in the build container) or, if that .so did not
    travel, the oracle port — timed on this host's cores on a bounded sample of the same workload."""
    import multiprocessing as mp
    import numpy as np
    from oracle.pyoracle import Oracle, ref_available
    o = Oracle()
    kind = "reference" if ref_available() else "port"
    cores = min(os.cpu_count() or 1, 16)
    if kind == "port":
        per_proc *= 8  # the flat restatement is roughly an order of magnitude faster per core
    total = per_proc * cores
    y = o.transmit_frames(cws[np.arange(total) % len(cws)], snr, first_seed=1)
    chunks = [(kind, Hm, y[i * per_proc:(i + 1) * per_proc], snr, max_iter) for i in range(cores)]
    ctx = mp.get_context("spawn")
    t0 = time.time()
    with ctx.Pool(cores) as pool:
        res = pool.map(cpu_baseline_worker, chunks)
    wall = time.time() - t0
    busy = max(r[0] for r in res)
    return {
        "value": total / busy, "unit": "frames/s", "cores": cores, "kind": kind,
        "sample": "%d frames (%d per process, %d single-threaded processes) of the same H05/AWGN workload at "
                  "%.1f dB, %s BeliefPropagationDecoder(%d) with its early exit; slowest process %.1f s, "
                  "pool wall %.1f s" % (total, per_proc, cores, snr,
                                       "reference" if kind == "reference" else "oracle port of", max_iter, busy, wall),
        "decoded_ok": sum(r[1] for r in res), "frames": total,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=1 << 20, help="frames per GPU per step (config 2: 1M)")
    ap.add_argument("--snr", type=float, default=-2.0)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--matrix", default=os.path.join(ROOT, "data", "H05.txt"))
    ap.add_argument("--lanes", type=int, default=0, help="lanes per frame (0 = library default)")
    ap.add_argument("--engine", choices=["auto", "fused", "streamed"], default="auto",
                    help="BP engine (default auto = fused LDS-resident for H05; streamed = messages in HBM)")
    ap.add_argument("--algo", choices=["bp", "minsum", "qpadmm"], default="bp")
    ap.add_argument("--alpha", type=float, default=1.95)
    ap.add_argument("--mu", type=float, default=0.5)
    ap.add_argument("--synthetic", type=int, nargs=4, metavar=("M", "N", "DV", "DC"), default=None,
                    help="use a seeded (dv,dc)-regular M x N code instead of --matrix (configs[4]: 5000 10000 3 6)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the early-exit side measurements")
    ap.add_argument("--cpu-frames-per-proc", type=int, default=1500)
    a = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import acg_alp_ldpc_amd as A
    from acg_alp_ldpc_amd._lib import McCfg, check, lib
    import ctypes as C

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if a.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with torch.distributed.run (one process per GPU)")
    if not A.device_available():
        raise SystemExit("bench.py needs a HIP device; the product has no CPU path")
    # one process per GPU.  ACG_BENCH_BACKEND=gloo (+ ranks sharing a GPU) is only for rehearsing the N>1 control
    # path on a one-GPU box; the driver's runs use nccl (= RCCL) with one GPU per rank.
    backend = os.environ.get("ACG_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if a.synthetic:
        m_, n_, dv_, dc_ = a.synthetic
        H = A.ParityCheckMatrix(A.regular_ldpc(m_, n_, dv_, dc_, seed=1))
        a.matrix = "synthetic_%dx%d_(%d,%d)" % (m_, n_, dv_, dc_)
        cws = np.zeros((1, n_), dtype=np.uint8)     # all-zero codeword (both decoders are symmetric)
    else:
        H = A.read_pcm(a.matrix)
        G, ok = H.get_orthogonal()
        assert ok
        cws = A.gen_random_codewords(G, 8192, 239239239)
    n, E, F = H.n, H.E, a.frames
    nw = (n + 31) // 32
    eng = {"auto": A.ENGINE_AUTO, "fused": A.ENGINE_FUSED, "streamed": A.ENGINE_STREAMED}[a.engine]

    def make(early_exit):
        if a.algo == "qpadmm":
            # fixed work for QP-ADMM = eps_stop 0 (the residual test never fires): every frame runs max_iter sweeps
            return A.QPADMMDecoder(a.alpha, a.mu, a.iters, 1e-5 if early_exit else 0.0, device=local_rank,
                                   lanes_per_frame=a.lanes)
        if a.algo == "minsum":
            return A.MinSumDecoder(a.iters, 0.75, early_exit=early_exit, device=local_rank, lanes_per_frame=a.lanes,
                                   engine=eng)
        return A.BeliefPropagationDecoder(a.iters, early_exit=early_exit, device=local_rank, lanes_per_frame=a.lanes,
                                          engine=eng)

    dec_fixed = make(False)
    dec_exit = make(True)

    y = torch.empty((F, n), dtype=torch.float32, device="cuda")
    bits = torch.zeros((F, nw), dtype=torch.int32, device="cuda")
    okf = torch.zeros(F, dtype=torch.uint8, device="cuda")
    its = torch.zeros(F, dtype=torch.int32, device="cuda")
    # a dedicated (non-null) HIP stream: kernels and the timing events are issued on the same stream
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)

    def gen_noise(snr):
        h, _ = dec_fixed.handle(H)
        cfg = McCfg()
        cfg.frames, cfg.first_frame, cfg.snr, cfg.seed, cfg.noise = F, rank * F, snr, 1, 0
        cfg.codewords, cfg.n_codewords = cws.ctypes.data, cws.shape[0]
        check(lib().acg_ldpc_awgn_dev(h, C.byref(cfg), y.data_ptr(), stream.cuda_stream))
        torch.cuda.synchronize()

    def run(dec, snr, steps, warmup):
        """-> (seconds for `steps` steps (max over ranks), mean kernel ms on this rank)"""
        for _ in range(warmup):
            dec.decode_batch_dev(H, y.data_ptr(), False, F, snr, bits.data_ptr(), okf.data_ptr(), its.data_ptr(),
                                 stream.cuda_stream)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        barrier()
        t0 = time.perf_counter()
        for s in range(steps):
            ev[s][0].record(stream)   # HIP events on the stream the kernel is launched on
            dec.decode_batch_dev(H, y.data_ptr(), False, F, snr, bits.data_ptr(), okf.data_ptr(), its.data_ptr(),
                                 stream.cuda_stream)
            ev[s][1].record(stream)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        kms = sum(e0.elapsed_time(e1) for e0, e1 in ev) / steps
        return dt, kms

    def quality():
        """FER / mean iterations of the batch currently in (bits, okf, its), all ranks"""
        idx = (torch.arange(F, device="cuda", dtype=torch.int64) + rank * F) % cws.shape[0]
        pad = np.zeros((cws.shape[0], nw * 32), dtype=np.uint8)
        pad[:, :n] = cws
        cwp = torch.from_numpy(np.packbits(pad, axis=1, bitorder="little").view(np.int32).copy()).cuda()
        good = (bits == cwp[idx]).all(dim=1) & (okf == 1)  # (QP-ADMM: ok is always 1; a wrong word is counted by the compare)
        v = torch.stack([good.sum(), okf.sum(), its.sum(), torch.tensor(F, device="cuda")]).to(torch.int64).to(red_dev)
        if world > 1:
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
        c, k, i, t = (int(x) for x in v.tolist())
        return {"fer": (t - c) / t, "undetected": k - c, "mean_iters": i / t, "frames": t}

    # ---- headline: fixed 50 iterations, SNR a.snr --------------------------------------------------------
    gen_noise(a.snr)
    dt, kms = run(dec_fixed, a.snr, a.steps, a.warmup)
    q_fixed = quality()
    value = world * F * a.steps / dt
    if a.algo == "qpadmm":
        sh = H.admm_shape()
        bpf = admm_bytes_per_frame(n, sh["n_con"], sh["n_var"], a.iters)
    else:
        bpf = bp_bytes_per_frame(n, E, a.iters)
    achieved = F * bpf / (kms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    engine_name = "streamed" if dec_fixed.layout(H)["lanes_per_frame"] == 1 else "fused"
    if os.path.exists(tpath):
        try:
            for tj in json.load(open(tpath)):
                if (tj.get("frames") == F and tj.get("iters") == a.iters and tj.get("engine") == engine_name
                        and tj.get("matrix") == os.path.basename(a.matrix) and tj.get("algo") == a.algo):
                    traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    # The fused engine is bound by VALU issue, not by HBM: price it against that too.  Instruction counts per launch
    # come from the committed rocprofv3 PMC pass of this exact configuration (they do not change with the clock);
    # issue cost 2 cycles per wave64 VALU op, 12 per transcendental (tools/microbench/valu_rates.hip), 1024 SIMDs.
    valu = None
    try:
        for pf in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
            if not pf.endswith("_summary.json"):
                continue
            pj = json.load(open(os.path.join(ROOT, "profiles", pf)))
            tj = pj.get("traffic", {})
            if (tj.get("frames") == F and tj.get("iters") == a.iters and tj.get("engine") == engine_name
                    and tj.get("matrix") == os.path.basename(a.matrix) and tj.get("algo") == a.algo):
                pm = pj["pmc_mean_per_launch"]
                nv, nt = pm["SQ_INSTS_VALU"], pm.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
                clk = pm["GRBM_GUI_ACTIVE"] / 8 / (pj["kernel_stats"][0]["avg_ms"] * 1e-3)
                busy = ((nv - nt) * 2 + nt * 12) / 1024
                valu = {"shader_clock_hz": clk, "source": "profiles/" + pf}
                if a.algo != "qpadmm":  # fp64 VALU ops issue at 4 cycles: the 2/12-cycle pricing only fits the fp32 kernels
                    valu.update({"issue_cycles_per_simd_per_launch": busy, "frac_of_valu_issue_capacity": busy / (clk * kms * 1e-3)})
                if "SQ_LDS_IDX_ACTIVE" in pm:  # LDS-array cycles (incl. bank conflicts) summed over the 256 CUs
                    valu["frac_of_lds_array_cycles"] = pm["SQ_LDS_IDX_ACTIVE"] / 256 / (clk * kms * 1e-3)
                    valu["lds_bank_conflict_share"] = pm.get("SQ_LDS_BANK_CONFLICT", 0.0) / pm["SQ_LDS_IDX_ACTIVE"]
    except Exception:
        valu = None
    out = {
        "metric": "decoded frames/sec (+ FER@SNR) for H05.txt 50-iter BP", "value": value, "unit": "frames/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": ("f64" if a.algo == "qpadmm" else "f32"), "data": "synthetic",
        "config": {"workload": "%s: %s (%dx%d, E=%d) %s, %d iterations FIXED (no early exit), "
                               "%d AWGN frames per GPU per step at Es/N0 %.1f dB, inputs/outputs resident in HBM"
                               % ("configs[4]" if a.synthetic else ("configs[2]" if a.algo == "qpadmm" else "configs[1]"), os.path.basename(a.matrix), H.m, n, E,
                                  {"bp": "sum-product BP", "minsum": "min-sum(0.75) BP",
                                   "qpadmm": "QP-ADMM(%g,%g) fp64" % (a.alpha, a.mu)}[a.algo], a.iters, F, a.snr),
                   "engine": "streamed (messages in HBM)" if dec_fixed.layout(H)["lanes_per_frame"] == 1
                             else "fused (messages in LDS)",
                   "frames_per_gpu": F, "snr_db": a.snr, "iters": a.iters, "early_exit": False,
                   "sharding": "frames [r*F,(r+1)*F) per rank, no collective on the data path",
                   "layout": dec_fixed.layout(H)},
        "fer": q_fixed["fer"], "undetected_errors": q_fixed["undetected"], "mean_exit_iter": q_fixed["mean_iters"],
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "model": ("streamed model (SURVEY 8d): %d B/frame x %d frames / %.3f ms mean kernel time; "
                               % (bpf, F, kms)) +
                              ("messages live in HBM: this IS the HBM figure" if dec_fixed.layout(H)["lanes_per_frame"] == 1
                               else "streamed-EQUIVALENT only: messages stay in LDS, so this is NOT HBM utilisation"),
                     "kernel_ms": kms, "bytes_per_frame": bpf, "valu_issue": valu},
    }

    # ---- reference semantics (early exit) at a.snr and +2 dB ---------------------------------------------
    if not a.no_extras:
        ee = {}
        for snr in (a.snr, 2.0):
            gen_noise(snr)
            dte, kmse = run(dec_exit, snr, max(3, a.steps // 2), 1)
            q = quality()
            st = max(3, a.steps // 2)
            bpf_e = (admm_bytes_per_frame(n, sh["n_con"], sh["n_var"], q["mean_iters"]) if a.algo == "qpadmm"
                     else bp_bytes_per_frame(n, E, q["mean_iters"]))
            ee["%+.1fdB" % snr] = {"value": world * F * st / dte, "unit": "frames/s", "ms_per_step": dte / st * 1e3,
                                    "kernel_ms": kmse, "fer": q["fer"], "mean_iters": q["mean_iters"],
                                    "roofline_frac_streamed_equiv": F * bpf_e / (kmse * 1e-3) / 1e9 / HBM_PEAK_GBS}
        out["early_exit"] = ee
        # north_star's named variant: normalised min-sum (NOT in the reference: parity unpinned)
        if a.algo == "bp" and not a.synthetic:
            ms = {}
            for early, tag in ((False, "fixed"), (True, "early_exit")):
                dms = A.MinSumDecoder(a.iters, 0.75, early_exit=early, device=local_rank, lanes_per_frame=a.lanes, engine=eng)
                gen_noise(a.snr)
                dtm, kmsm = run(dms, a.snr, max(3, a.steps // 2), 1)
                q = quality()
                st = max(3, a.steps // 2)
                ms[tag] = {"value": world * F * st / dtm, "unit": "frames/s", "kernel_ms": kmsm, "fer": q["fer"],
                           "mean_iters": q["mean_iters"], "snr_db": a.snr}
                dms.close()
            out["minsum_0.75"] = ms
            # configs[2]: QP-ADMM(alpha, mu) fp64, 100 sweeps, on the first quarter of the same frames (the run() helper
            # decodes F frames, so the decoder is driven directly here); eps_stop 0 = every frame runs all sweeps
            Fq = max(1, F // 4)
            qa = {}
            for eps, tag in ((0.0, "fixed"), (1e-5, "residual_exit")):
                dq = A.QPADMMDecoder(a.alpha, a.mu, 100, eps, device=local_rank)
                gen_noise(a.snr)
                st = 3
                for it_ in range(st + 1):
                    if it_ == 1:
                        barrier()
                        t0 = time.perf_counter()
                    dq.decode_batch_dev(H, y.data_ptr(), False, Fq, a.snr, bits.data_ptr(), okf.data_ptr(), its.data_ptr(),
                                        stream.cuda_stream)
                barrier()
                dtq = time.perf_counter() - t0
                if world > 1:
                    tq = torch.tensor([dtq], dtype=torch.float64, device=red_dev)
                    dist.all_reduce(tq, op=dist.ReduceOp.MAX)
                    dtq = float(tq.item())
                qa[tag] = {"value": world * Fq * st / dtq, "unit": "frames/s", "ms_per_step": dtq / st * 1e3,
                           "frames_per_gpu": Fq, "mean_sweeps": float(its[:Fq].double().mean().item()), "snr_db": a.snr,
                           "dtype": "f64", "layout": dq.layout(H)}
                dq.close()
            out["qpadmm_%g_%g_100" % (a.alpha, a.mu)] = qa

    # ---- CPU baseline (rank 0, N = 1 only) -------------------------------------------------------------
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(H.dense(), cws, a.snr, a.iters, a.cpu_frames_per_proc)
    elif world == 1:
        out["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
