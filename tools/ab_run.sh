#!/bin/bash
# Developer A/B helper (GPU box): QP-ADMM fixed-100 bench + Monte-Carlo rate for every library variant given.
#   tools/ab_run.sh NAME...     (variants built by tools/ab_build.sh; "main" = the regular library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for name in "$@"; do
  if [ "$name" = main ]; then unset ACG_LDPC_LIB; else export ACG_LDPC_LIB=$PWD/acg_alp_ldpc_amd/lib/variants/libacg_$name.so; fi
  timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --algo qpadmm --iters 100 --frames 262144 ${BENCH_ARGS} > gpurun_out/ab_$name.log 2>&1
  python - "$name" <<'PY'
import json, sys
name = sys.argv[1]
ok = False
for l in open("gpurun_out/ab_%s.log" % name):
    if l.startswith("{"):
        j = json.loads(l); ok = True
        print("%-10s fixed %.3f M/s kernel %.2f ms fer %.4f layout %s" % (name, j["value"] / 1e6, j["roofline"]["kernel_ms"], j["fer"], j["config"]["layout"]))
if not ok:
    print(name, open("gpurun_out/ab_%s.log" % name).read()[-800:])
PY
  timeout -k 10 200 python tools/mc_rate.py --algo qpadmm ${MC_ARGS} 2>&1 | grep "dB" | sed "s/^/  $name /"
done
