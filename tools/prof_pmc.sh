#!/bin/bash
# rocprofv3 PMC passes (each counter group in its own run, no trace domains besides kernel-trace).
# usage: tools/prof_pmc.sh <tag> <counters...>   e.g. tools/prof_pmc.sh fetch FETCH_SIZE
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
tag=$1; shift
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/prof_pmc_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras ${BENCH_ARGS} > gpurun_out/prof_pmc_$tag.log 2>&1
echo "rc=$?" >> gpurun_out/prof_pmc_$tag.log
tail -2 gpurun_out/prof_pmc_$tag.log
f=$(find gpurun_out/prof_pmc_$tag -name "*counter_collection.csv" | head -1)
echo "== $f"
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "bp_" not in k and "admm" not in k: continue
    for c, v in d.items():
        print(k, c, "n=%d mean=%.6g" % (len(v), sum(v) / len(v)))
PY
