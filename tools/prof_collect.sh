#!/bin/bash
# Collects the evidence bench.py's roofline object refers to (run on the GPU box):
#   1. rocprofv3 --kernel-trace --stats of the default bench command
#   2. PMC passes, each alone: FETCH_SIZE, WRITE_SIZE(+GRBM_GUI_ACTIVE), SQ issue counters
# and condenses them into gpurun_out/profile_summary.{md,json} (copy into profiles/).
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-extras $*"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py $ARGS > gpurun_out/prof_stats.log 2>&1 || exit 1
for spec in "fetch:FETCH_SIZE" "write:WRITE_SIZE GRBM_GUI_ACTIVE" "sq1:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "sq2:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32"; do
  tag=${spec%%:*}; ctr=${spec#*:}
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/prof_pmc_$tag -- python3 bench.py $ARGS > gpurun_out/prof_pmc_$tag.log 2>&1 || exit 1
done
python3 tools/prof_summarize.py gpurun_out "$ARGS"
