#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench line (GPU box). Output under gpurun_out/prof_stats.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras "$@" > gpurun_out/prof_stats.log 2>&1
echo "rc=$?" >> gpurun_out/prof_stats.log
tail -3 gpurun_out/prof_stats.log
f=$(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1)
echo "== $f"; head -8 "$f"
