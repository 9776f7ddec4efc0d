#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command and of the headline alone (GPU box); raw output under
# gpurun_out/$1/, summary written by tools/prof_summarize.py.      tools/prof_stats.sh r03
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/$tag
rm -rf gpurun_out/$tag/prof_stats gpurun_out/$tag/prof_stats_hl
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof_stats -- python3 bench.py --no-pmc --no-cpu-baseline --steps 5 --warmup 1 --detail-out gpurun_out/$tag/prof_stats_detail.json > gpurun_out/$tag/prof_stats.log 2>&1
echo "rc=$?" >> gpurun_out/$tag/prof_stats.log
f=$(find gpurun_out/$tag/prof_stats -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/$tag/bench_default_kernel_stats.csv
head -24 "$f" | cut -c1-200
# the headline alone: every launch of its kernel is then a fixed-50, 1M-frame launch
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof_stats_hl -- python3 bench.py --no-extras --no-pmc --no-cpu-baseline --steps 10 --warmup 2 --detail-out gpurun_out/$tag/prof_stats_hl_detail.json > gpurun_out/$tag/prof_stats_hl.log 2>&1
echo "rc=$?" >> gpurun_out/$tag/prof_stats_hl.log
f=$(find gpurun_out/$tag/prof_stats_hl -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/$tag/bench_headline_kernel_stats.csv
head -4 "$f" | cut -c1-200
