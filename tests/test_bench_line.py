"""bench.py's output contract and launchers, on CPU.

* The LAST stdout line of bench.py must fit the driver's 8 KB tail with room to spare: `compact_line` (a pure function of the
  detail object) stays under 4096 bytes and round-trips through json — built here from the committed round-2 detail object
  (profiles/r02_bench_default.json, 24 KB: the line that overflowed the driver's record in round 2).
* The two launchers of the N > 1 path (one process per GPU over torch.distributed/gloo; one process driving N devices,
  --inproc) must produce the SAME line shape and the SAME merged counters for the same shards.  No GPU here, so the shard
  function behind bench.mc_leg is the oracle (test infrastructure) fed with the global frame range the C ABI would have been
  asked for; shard order, the gloo reductions, the per-shard gather and the line builder are bench.py's own code.
Reference for the sharding and merge: experiment.h:70-78,125-139."""
import json
import os
import sys
import types

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _detail_r02():
    return json.load(open(os.path.join(ROOT, "profiles", "r02_bench_default.json")))


def test_compact_line_fits_and_round_trips():
    import bench
    d = _detail_r02()
    assert len(json.dumps(d)) > 20000          # the object that did not fit
    c = bench.compact_line(d)
    line = json.dumps(c)
    assert len(line) < bench.COMPACT_LIMIT == 4096
    back = json.loads(line)
    assert back == c
    # the contract's keys, with the numbers of the detail object
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in back, k
    assert abs(back["value"] / d["value"] - 1) < 1e-5 and abs(back["ms_per_step"] / d["ms_per_step"] - 1) < 1e-4
    assert back["config"]["workload"].startswith("configs[1]") and back["config"]["frames_per_gpu"] == 1 << 20
    for k in ("bound", "frac", "achieved", "peak", "unit", "traffic", "kernel_ms", "streamed_equiv_frac", "frac_by_op_class"):
        assert k in back["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample", "frames"):
        assert k in back["cpu_baseline"] and k in back["cpu_baseline_qpadmm"], k
    legs = back["legs"]
    # one triple per side leg: early exit, Monte-Carlo (SURVEY 8(d)'s definitional metric), streamed, configs[2], configs[4]
    for k in ("early_exit_-2.0dB", "mc_bp50_-2.0dB", "streamed_h05_sum_product", "configs[2]_qpadmm_fixed", "configs[2]_qpadmm_exit",
              "configs[4]_fused_block_minsum (parity unpinned)", "configs[4]_fused_pair_f16_minsum (parity unpinned)",
              "configs[4]_streamed_minsum (parity unpinned)"):
        assert set(legs[k]) >= {"value", "frac"}, k
    for k in ("early_exit_-2.0dB", "streamed_h05_sum_product", "configs[2]_qpadmm_fixed", "configs[4]_streamed_minsum (parity unpinned)"):
        assert legs[k]["frac"] is not None and legs[k]["bound"], k          # the legs with a counter-backed roofline name their bound
    # every min-sum figure carries the label (min-sum is not in the reference)
    assert all("parity unpinned" in k for k in legs if "minsum" in k)


def test_compact_line_sheds_fields_rather_than_overflow():
    import bench
    d = _detail_r02()
    d["per_shard_ms_per_step"] = [53.123456] * 8
    d["pmc_note"] = "x" * 500
    for i in range(40):   # far more legs than any run produces
        d["configs[4]"]["extra_leg_%d_minsum" % i] = {"value": 1.0e6 + i, "fer": 0.0, "roofline": {"frac": 0.5, "bound": "hbm"}}
    line = json.dumps(bench.compact_line(d))
    assert len(line) < 4096
    c = json.loads(line)
    assert c["value"] and c["roofline"]["frac"] and c["cpu_baseline"]["value"]      # the headline never goes
    assert len(c["per_shard_ms_per_step"]) == 8                                      # 8 shards stay visible


def test_emit_prints_the_compact_line_last(tmp_path, capsys):
    import bench
    d = _detail_r02()
    bench.emit(d, str(tmp_path / "bench_detail.json"))
    out = capsys.readouterr().out
    lines = out.rstrip("\n").split("\n")
    assert lines[-2].startswith("BENCH_DETAIL ") and json.loads(lines[-2][len("BENCH_DETAIL "):]) == d
    assert len(lines[-1]) < 4096 and json.loads(lines[-1])["value"] == bench.compact_line(d)["value"]
    assert out.endswith("\n") and json.load(open(tmp_path / "bench_detail.json")) == d


def test_roofline_frac_is_the_guaranteed_figure():
    """roofline.frac = counters x the guide's cycle constants (2 per VALU instruction, 8 per transcendental); the
    operation-class estimate is a side field and never replaces it"""
    import bench
    d = _detail_r02()
    c = dict(d["pmc"]["items"]["bp_fused"])
    r = bench.roofline_fused(c, "test", d["roofline"]["kernel_ms"], 1 << 20, 745155)
    nv, nt = c["SQ_INSTS_VALU"], c["SQ_INSTS_VALU_TRANS_F32"]
    want = ((nv - nt) * 2 + nt * 8) / (d["roofline"]["kernel_ms"] * 1e-3) / (1024 * 2.4e9)
    assert abs(r["frac"] - want) < 1e-9 and 0.74 < r["frac"] < 0.77
    assert r["frac_by_op_class"] > r["frac"] and r["bound"] == "valu_issue"
    h = bench.roofline_hbm(None, None, 105.0, 1 << 20, 745155, working_set=768 * 296 * 1024)
    assert h["bound"] == "fabric_l2" and h["infinity_cache_resident"] is True
    h = bench.roofline_hbm(None, None, 158.0, 32768, 26041250, working_set=5324800000)
    assert h["bound"] == "hbm" and 0.6 < h["frac"] < 0.7


def test_every_probed_item_has_a_kernel_pattern():
    """(round 3 lost the QP-ADMM counters for two runs to a comment in the middle of the table)"""
    import bench
    assert set(bench.PROBE_KERNEL) == set(bench.PROBE_ITEMS)
    for item, (pat, pos) in bench.PROBE_KERNEL.items():
        assert (isinstance(pat, str) and pat) or (isinstance(pat, tuple) and all(pat)), item
        assert isinstance(pos, int)


def test_probe_patterns_name_kernels_of_the_built_library():
    """every kernel bench.py asks rocprofv3 about exists under that (demangled) name in libacg_ldpc_hip.so — a template
    parameter added to a kernel silently emptied two counter sets in round 3"""
    import bench
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import valu_mix
    if not os.path.exists(os.path.join(valu_mix.LLVM, "llvm-objdump")):
        pytest.skip("llvm-objdump not available")
    import acg_alp_ldpc_amd as A
    A.build()
    names = list(valu_mix.disassemble(valu_mix.default_lib()))
    for item, (pat, _) in bench.PROBE_KERNEL.items():
        pats = (pat,) if isinstance(pat, str) else pat
        assert any(q in k for q in pats for k in names), (item, pats)


def test_usable_cores_reads_affinity_not_cpu_count(monkeypatch):
    import bench
    monkeypatch.setattr(os, "cpu_count", lambda: 256)
    used, info = bench.usable_cores(None)
    assert info["cores_visible"] == 256 and used == info["cores_used"] <= len(os.sched_getaffinity(0))


# ---------------------------------------------------------------------------------------------------------------------------
def _oracle_shard(o, Hm, cws, snr, lo, cnt, max_iter):
    n = Hm.shape[1]
    idx = np.arange(lo, lo + cnt) % len(cws)
    y = np.stack([o.transmit(lo + i + 1, snr, cws[idx[i]]) for i in range(cnt)]) if cnt else np.zeros((0, n))
    bits, ok, iters = o.bp_decode(Hm, y, snr, max_iter, threads=2)
    sent = cws[idx]
    good = (ok == 1) & (bits == sent).all(axis=1)
    ham = ((sent == 0) & (y <= 0)).sum(axis=1) + ((sent == 1) & (y > 0)).sum(axis=1)
    return np.array([good.sum(), ((ok == 1) & ~good).sum(), cnt, ham.sum(), ham[good].sum(), ham[~good].sum(), iters.sum()], np.int64)


class _FakeDecoder:
    def __init__(self, dev):
        self.dev = dev

    def close(self):
        pass


def _mc_leg_with_oracle(world_args, frames, snr):
    """bench.mc_leg with the C-ABI call replaced by the oracle on exactly the global frames the ABI would have got"""
    import acg_alp_ldpc_amd as A
    import acg_alp_ldpc_amd.experiment as E
    import bench
    from oracle.pyoracle import Oracle
    o = Oracle()
    Hm = o.read_pcm(os.path.join(ROOT, "data", "H.txt"))
    G, _ = o.get_orthogonal(Hm)
    cws = o.gen_codewords(G, 239239239, 97)

    def fake_run(decoder, codewords, H, snr, frames=None, first_frame=0, noise="host", seed=1):
        return E.ExperimentResult.from_vector(_oracle_shard(o, Hm, codewords, snr, int(first_frame), int(frames), 20), kernel_ms=1.0)
    old = A.run_experiment
    A.run_experiment = fake_run
    try:
        rig = bench.Rig(world_args, cpu=True)
        leg = bench.mc_leg(rig, None, cws, _FakeDecoder, snr, frames, 1, noise="host")
        detail = {"metric": "m", "value": leg["value"], "unit": "frames/s", "n_gpus": rig.nshards, "steps": 1, "warmup": 0,
                  "ms_per_step": leg["ms_per_step"], "dtype": "f32", "config": {"workload": "w", "launcher": rig.launcher},
                  "per_shard_ms_per_step": leg["per_shard_ms_per_step"], "roofline": {}, "monte_carlo": {"bp": leg}}
        line = bench.compact_line(detail)
        rank = rig.rank
        rig.close()
    finally:
        A.run_experiment = old
    return rank, leg, line


def _gloo_worker(rank, world, port, frames, snr, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                       "LOCAL_RANK": str(rank)})
    a = types.SimpleNamespace(gpus=world, inproc=0)
    r, leg, line = _mc_leg_with_oracle(a, frames, snr)
    if r == 0:
        json.dump({"leg": leg, "line": line}, open(out_path, "w"))


def test_gloo_line_and_inproc_line_have_identical_keys_and_counters(tmp_path):
    frames, snr = 75, 0.5      # frames per shard
    out = str(tmp_path / "gloo.json")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_gloo_worker, args=(2, port, frames, snr, out), nprocs=2, join=True)
    g = json.load(open(out))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    _, leg_i, line_i = _mc_leg_with_oracle(types.SimpleNamespace(gpus=1, inproc=2), frames, snr)
    assert set(g["leg"]) == set(leg_i) and set(g["line"]) == set(line_i)
    assert g["leg"]["counters"] == leg_i["counters"]                      # merged over the two shards, either launcher
    assert g["line"]["n_gpus"] == line_i["n_gpus"] == 2
    assert len(g["line"]["per_shard_ms_per_step"]) == len(line_i["per_shard_ms_per_step"]) == 2
    assert all(x > 0 for x in g["line"]["per_shard_ms_per_step"] + line_i["per_shard_ms_per_step"])   # a straggler would show here
    assert set(g["line"]["legs"]) == set(line_i["legs"]) == {"mc_bp"}
    # and both equal the single-process run over the union of the shards (global frames 0 .. 2*frames)
    from oracle.pyoracle import Oracle
    o = Oracle()
    Hm = o.read_pcm(os.path.join(ROOT, "data", "H.txt"))
    G, _ = o.get_orthogonal(Hm)
    cws = o.gen_codewords(G, 239239239, 97)
    assert list(_oracle_shard(o, Hm, cws, snr, 0, 2 * frames, 20)) == leg_i["counters"]
