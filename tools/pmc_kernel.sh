#!/bin/bash
# Developer tool (GPU box): one rocprofv3 counter pass over tools/rate.py; prints per-kernel means.
#   tools/pmc_kernel.sh <tag> "<counters>" <rate.py args...>
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=$1; ctr=$2; shift 2
out=gpurun_out/pmc_$tag
rm -rf $out
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -- python3 tools/rate.py --steps 2 "$@" > $out.log 2>&1
tail -1 $out.log
python3 - "$out" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        per[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), d in per.items():
        for c, v in d.items():
            agg[k][c].append(v)
for k, d in agg.items():
    if not any(x in k for x in ("bp_", "admm")):
        continue
    print(k[:100])
    for c, v in sorted(d.items()):
        print("   %-28s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
PY
