#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
for g in 0 1; do MC_GEMM_ONEWAVE128=$g timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-vae > gpurun_out/r03_bench_onewave$g.json 2>/dev/null; python - <<PY
import json
l=[x for x in open("gpurun_out/r03_bench_onewave$g.json") if x.startswith("{")]
j=json.loads(l[-1]); print("onewave128=$g videos/min", j["value"], "eager", j["eager"]["videos_per_min"])
PY
done
