#!/bin/bash
# quick A/B on the GPU box: smoke + bench for several lanes-per-frame settings
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -1
for L in ${LANES:-64 32}; do
  timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --lanes $L "$@" > gpurun_out/bench_L$L.log 2>&1
  python - "$L" <<'PY'
import json, sys
L = sys.argv[1]
ok = False
for l in open("gpurun_out/bench_L%s.log" % L):
    if l.startswith("{"):
        j = json.loads(l); ok = True
        print("L=%s fixed50 %.3f M/s (%.1f ms, frac %.3f) ee %s fer %.4f layout %s" % (
            L, j["value"] / 1e6, j["roofline"]["kernel_ms"], j["roofline"]["frac"],
            {k: round(v["value"] / 1e6, 1) for k, v in j.get("early_exit", {}).items()}, j["fer"], j["config"]["layout"]))
if not ok:
    print(open("gpurun_out/bench_L%s.log" % L).read()[-1500:])
PY
done
