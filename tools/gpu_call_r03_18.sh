#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/r03_attn_cross_ab.jsonl
for r in 1 2; do MC_ATTN_RING=$r MC_ATTN_TAG="ring=$r" timeout 300 python tools/attn_bench.py 2>/dev/null | grep '"x[012]"' >> gpurun_out/r03_attn_cross_ab.jsonl; done
cat gpurun_out/r03_attn_cross_ab.jsonl
for n in 2 4; do timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-vae --inflight $n > gpurun_out/r03_bench_inflight$n.json 2>/dev/null; python - <<PY
import json
l=[x for x in open("gpurun_out/r03_bench_inflight$n.json") if x.startswith("{")]
j=json.loads(l[-1]); print("inflight $n videos/min", j["value"])
PY
done
