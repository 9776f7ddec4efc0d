#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command (GPU box); summary goes to gpurun_out/r02/prof_stats_*.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r02
rm -rf gpurun_out/r02/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/prof_stats -- python3 bench.py --no-pmc --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/r02/prof_stats.log 2>&1
echo "rc=$?" >> gpurun_out/r02/prof_stats.log
f=$(find gpurun_out/r02/prof_stats -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r02/bench_default_kernel_stats.csv
head -20 "$f" | cut -c1-200
# the headline alone: every launch of its kernel is then a fixed-50, 1M-frame launch
rm -rf gpurun_out/r02/prof_stats_hl
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/prof_stats_hl -- python3 bench.py --no-extras --no-pmc --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/r02/prof_stats_hl.log 2>&1
echo "rc=$?" >> gpurun_out/r02/prof_stats_hl.log
f=$(find gpurun_out/r02/prof_stats_hl -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r02/bench_headline_kernel_stats.csv
head -4 "$f" | cut -c1-200
