#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
for g in 1 3 4; do timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-vae --gemm-lanes $g > gpurun_out/r03_bench_gemmlanes$g.json 2>/dev/null; python - <<PY
import json
l=[x for x in open("gpurun_out/r03_bench_gemmlanes$g.json") if x.startswith("{")]
j=json.loads(l[-1]); print("gemm-lanes $g videos/min", j["value"])
PY
done
