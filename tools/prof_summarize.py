#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/prof_collect.sh into profile_summary.{md,json}."""
import collections
import csv
import glob
import json
import os
import sys

out_dir = sys.argv[1]
args = sys.argv[2] if len(sys.argv) > 2 else ""


def find(pat):
    r = sorted(glob.glob(os.path.join(out_dir, pat), recursive=True), key=os.path.getmtime)
    return r[-1] if r else None


summary = {"bench_args": args}
stats = find("prof_stats/**/*kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
main = [r for r in rows if any(k in r["Name"] for k in ("_fused_kernel", "_streamed_kernel", "_block_kernel"))]
summary["kernel_stats"] = [{"name": r["Name"][:90], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                            "min_ms": float(r["MinNs"]) / 1e6, "max_ms": float(r["MaxNs"]) / 1e6,
                            "pct": float(r["Percentage"])} for r in rows[:6]]
kname = main[0]["Name"]
pmc = collections.defaultdict(list)
for tag in ("fetch", "write", "sq1", "sq2"):
    f = find("prof_pmc_%s/**/*counter_collection.csv" % tag)
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"] == kname:
            pmc[r["Counter_Name"]].append(float(r["Counter_Value"]))
summary["pmc_mean_per_launch"] = {k: sum(v) / len(v) for k, v in pmc.items()}
p = summary["pmc_mean_per_launch"]
line = None
for l in open(os.path.join(out_dir, "prof_stats.log")):
    if l.startswith("{"):
        line = json.loads(l)
summary["bench_line"] = line
if "FETCH_SIZE" in p and "WRITE_SIZE" in p and line:
    F = line["config"]["frames_per_gpu"]
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  gfx950: FETCH_SIZE counts 64 B per 128-B request, i.e.
    # exactly half of a coalesced streaming read (MI355X_MICROARCH.md §HBM) -> x2.  Calibration on this kernel:
    # the only large read is the input batch, F*n*4 bytes, read exactly once.
    fetch = p["FETCH_SIZE"] * 1024 * 2
    write = p["WRITE_SIZE"] * 1024
    import re
    n = int(re.search(r"\((\d+)x(\d+), E=", line["config"]["workload"]).group(2))
    summary["traffic"] = {"frames": F, "iters": line["config"]["iters"],
                          "matrix": line["config"]["workload"].split(": ")[1].split(" ")[0],
                          "engine": "streamed" if line["config"]["layout"]["lanes_per_frame"] == 1 else "fused",
                          "algo": "minsum" if "min-sum" in line["config"]["workload"] else ("qpadmm" if "QP-ADMM" in line["config"]["workload"] else "bp"),
                          "fetch_bytes_corrected": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
                          "compulsory_input_bytes": F * n * 4, "kernel_ms": line["roofline"]["kernel_ms"],
                          "hbm_GBps_measured": (fetch + write) / (line["roofline"]["kernel_ms"] * 1e-3) / 1e9,
                          "algorithmic_streamed_bytes": line["roofline"]["bytes_per_frame"] * F,
                          "note": "FETCH_SIZE x2 (gfx950 half-count of wide reads), WRITE_SIZE as is; separate --pmc passes"}
json.dump(summary, open(os.path.join(out_dir, "profile_summary.json"), "w"), indent=1)
with open(os.path.join(out_dir, "profile_summary.md"), "w") as f:
    f.write("# rocprofv3 summary\n\nbench args: `%s`\n\n## kernel-trace --stats (top rows)\n\n" % args)
    f.write("| kernel | calls | avg ms | min ms | max ms | % |\n|---|---|---|---|---|---|\n")
    for r in summary["kernel_stats"]:
        f.write("| `%s` | %d | %.3f | %.3f | %.3f | %.2f |\n" % (r["name"], r["calls"], r["avg_ms"], r["min_ms"], r["max_ms"], r["pct"]))
    f.write("\n## PMC, mean per launch of the decode kernel (separate passes)\n\n| counter | value |\n|---|---|\n")
    for k, v in sorted(p.items()):
        f.write("| %s | %.6g |\n" % (k, v))
    if "traffic" in summary:
        t = summary["traffic"]
        f.write("\n## HBM traffic per launch\n\n")
        for k, v in t.items():
            f.write("* %s: %s\n" % (k, v))
    if line:
        f.write("\n## bench line of the profiled run\n\n```\n%s\n```\n" % json.dumps(line))
print(open(os.path.join(out_dir, "profile_summary.md")).read())
