#!/usr/bin/env python3
"""Condense one bench.py run (its detail object incl. the counter passes it took itself: bench_detail.json, or a log holding
the "BENCH_DETAIL " line) and the rocprofv3 --kernel-trace --stats CSV of the same command into profiles/rNN_summary.md + the
raw files.   tools/prof_summarize.py <bench_detail.json | bench log> <kernel_stats.csv> <tag>"""
import csv
import json
import os
import shutil
import sys

log, stats, tag = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if log.endswith(".json"):
    j = json.load(open(log))
else:
    line = [l for l in open(log) if l.startswith("BENCH_DETAIL ")][-1]
    j = json.loads(line[len("BENCH_DETAIL "):])
out = os.path.join(root, "profiles")
json.dump(j, open(os.path.join(out, "%s_bench_default.json" % tag), "w"), indent=1)
shutil.copy(stats, os.path.join(out, "%s_bench_default_kernel_stats.csv" % tag))
if j.get("pmc", {}).get("items"):
    json.dump({"source": "bench.py run summarised in profiles/%s_summary.md" % tag, "csrc_sha": j["pmc"]["csrc_sha"],
               "frames": j["config"]["frames_per_gpu"], "c5_frames": j["configs[4]"]["fused_block_minsum"]["frames_per_gpu"],
               "items": j["pmc"]["items"]}, open(os.path.join(out, "pmc_%s.json" % tag), "w"), indent=1)
rows = list(csv.DictReader(open(stats)))


def rl(r):
    if not r:
        return "—"
    f = r.get("frac")
    s = "%s %.3f" % (r.get("bound"), f) if f is not None else "%s n/a" % r.get("bound")
    for k, lab in (("valu_issue_frac", "VALU"), ("lds_array_frac", "LDS"), ("lds_bank_conflict_share", "conflicts"), ("wave_wait_share", "parked"),
                   ("traffic_over_algorithmic", "traffic/alg")):
        if r.get(k) is not None:
            s += ", %s %.3f" % (lab, r[k])
    return s


with open(os.path.join(out, "%s_summary.md" % tag), "w") as f:
    f.write("# %s — one `python bench.py` run on one MI355X, and `rocprofv3 --kernel-trace --stats` of the same command\n\n" % tag)
    f.write("Counters: taken by bench.py itself (`rocprofv3 --kernel-trace --pmc …` over `bench.py --pmc-probe`, one counter group per pass: %s); "
            "kernel sources `csrc_sha` %s.  Raw: `%s_bench_default.json`, `pmc_%s.json`, `%s_bench_default_kernel_stats.csv`.\n\n"
            % (", ".join(j["pmc"]["passes"]) if j.get("pmc") else "none", j.get("pmc", {}).get("csrc_sha"), tag, tag, tag))
    f.write("| measurement | frames/s | kernel ms | roofline (utilisation of the binding resource) |\n|---|---|---|---|\n")
    f.write("| **headline**: %s | %.4g | %.3f | %s |\n" % (j["config"]["workload"], j["value"], j["roofline"]["kernel_ms"], rl(j["roofline"])))
    for k, v in j.get("early_exit", {}).items():
        f.write("| early exit %s (FER %.4g, mean sweeps %.2f) | %.4g | %.3f | %s |\n" % (k, v["fer"], v["mean_iters"], v["value"], v["kernel_ms"], rl(v.get("roofline"))))
    for k, v in j.get("minsum_0.75", {}).items():
        if isinstance(v, dict):
            f.write("| min-sum(0.75) %s (parity unpinned; FER %.4g, mean iterations %.2f) | %.4g | %.3f | %s |\n"
                    % (k, v["fer"], v["mean_iters"], v["value"], v["kernel_ms"], rl(v.get("roofline"))))
    for k, v in j.get("streamed", {}).items():
        if isinstance(v, dict):
            f.write("| streamed engine, %s | %.4g | %.3f | %s; %s |\n" % (k, v["value"], v["kernel_ms"], rl(v["roofline"]), v["roofline"].get("note")))
    c2 = j.get("configs[2]", {})
    for k in ("fixed_100_sweeps", "residual_exit_1e-5"):
        if k in c2:
            f.write("| configs[2] QP-ADMM %s (FER %.4g, mean sweeps %.2f) | %.4g | %.3f | %s |\n"
                    % (k, c2[k]["fer"], c2[k]["mean_iters"], c2[k]["value"], c2[k]["kernel_ms"], rl(c2[k].get("roofline"))))
    for k, v in j.get("monte_carlo", {}).items():
        if isinstance(v, dict):
            f.write("| Monte-Carlo loop %s (FER %.4g) | %.4g | %.3f | %s |\n" % (k, v["fer"], v["value"], v.get("kernel_ms", 0), rl(v.get("roofline"))))
    for k, v in j.get("configs[4]", {}).items():
        if isinstance(v, dict):
            f.write("| configs[4] %s | %.4g | %.3f | %s |\n" % (k, v["value"], v["kernel_ms"], rl(v.get("roofline"))))
    for k in ("cpu_baseline", "cpu_baseline_qpadmm"):
        if j.get(k):
            f.write("| %s: %s | %.4g | — | %d cores, kind %s |\n" % (k, j[k]["sample"], j[k]["value"], j[k]["cores"], j[k]["kind"]))
    f.write("\n## rocprofv3 --kernel-trace --stats (top rows)\n\n| kernel | calls | avg ms | min ms | max ms | %% |\n|---|---|---|---|---|---|\n")
    for r in rows[:14]:
        f.write("| `%s` | %s | %.3f | %.3f | %.3f | %s |\n" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6,
                                                          float(r["MaxNs"]) / 1e6, r["Percentage"]))
    # the headline alone (`bench.py --no-extras`, the second run of tools/prof_stats.sh): every launch of its kernel is a
    # 50-sweep, 1M-frame launch, so the average duration here is the one to hold against `ms_per_step` / `roofline.kernel_ms`
    hl = os.path.join(os.path.dirname(stats), "bench_headline_kernel_stats.csv")
    if os.path.exists(hl):
        shutil.copy(hl, os.path.join(out, "%s_bench_headline_kernel_stats.csv" % tag))
        hrows = list(csv.DictReader(open(hl)))
        f.write("\n## rocprofv3 --kernel-trace --stats of `bench.py --no-extras` (headline launches only)\n\n| kernel | calls | avg ms | min ms | max ms |\n|---|---|---|---|---|\n")
        for r in hrows[:3]:
            f.write("| `%s` | %s | %.3f | %.3f | %.3f |\n" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
        f.write("\nbench.py, same sources: `ms_per_step` %.3f, `roofline.kernel_ms` %.3f (HIP events on the launch stream).\n" % (j["ms_per_step"], j["roofline"]["kernel_ms"]))
    if j.get("pmc", {}).get("items"):
        f.write("\n## counters, mean per launch\n\n")
        for item, c in j["pmc"]["items"].items():
            f.write("* **%s**: " % item + ", ".join(("%s %.6g" % (k, v)) if not isinstance(v, list) else ("%s [%s]" % (k, ", ".join("%.6g" % x for x in v)))
                                                      for k, v in sorted(c.items()) if not k.endswith("_per_dispatch") or k.startswith("GRBM")) + "\n")
print(open(os.path.join(out, "%s_summary.md" % tag)).read()[:3000])
