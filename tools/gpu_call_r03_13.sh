#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -k "attention" 2>&1 | tail -3
: > gpurun_out/r03_attn_ring_xcd.jsonl
for x in 0 1; do MC_ATTN_XCD=$x MC_ATTN_TAG="ring,xcd=$x" timeout 300 python tools/attn_bench.py --fwd-only >> gpurun_out/r03_attn_ring_xcd.jsonl 2>/dev/null; done
grep '"l0"\|"l1"' gpurun_out/r03_attn_ring_xcd.jsonl
timeout 900 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-vae > gpurun_out/r03_bench_attn_ring.json 2>gpurun_out/r03_bench_attn_ring.err
python - <<PY
import json
l=[x for x in open("gpurun_out/r03_bench_attn_ring.json") if x.startswith("{")]
j=json.loads(l[-1]); print("videos/min", j["value"], "ms/video", j["ms_per_step"], "e2e", j["roofline"])
for k,v in j.get("roofline_by_kernel",{}).items():
    if k.startswith("attn"): print("  ",k, round(v["avg_launch_us"],1), round(v["frac"],3), round(v["share_of_probe_video"],4))
PY
timeout 900 python -m pytest tests/test_fullsize_parity.py -m gpu -q -x 2>&1 | tail -3
