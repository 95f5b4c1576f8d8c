#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -k "attention" 2>&1 | tail -3
for x in 0 1; do MC_ATTN_XCD=$x timeout 300 python tools/attn_bench.py; done > gpurun_out/r03_attn_xcd_ab.jsonl 2>gpurun_out/r03_attn_xcd_ab.err
cat gpurun_out/r03_attn_xcd_ab.jsonl; tail -3 gpurun_out/r03_attn_xcd_ab.err
for x in 0 1; do MC_ATTN_XCD=$x timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-vae > gpurun_out/r03_bench_attn_xcd$x.json 2>gpurun_out/r03_bench_attn_xcd$x.err; python - <<PY
import json
l=[x for x in open("gpurun_out/r03_bench_attn_xcd$x.json") if x.startswith("{")]
j=json.loads(l[-1]); print("xcd=$x", j["value"], j["ms_per_step"])
for k,v in j.get("roofline_by_kernel",{}).items():
    if k.startswith("attn"): print("  ",k, round(v["avg_launch_us"],1), round(v["frac"],3), round(v["share_of_probe_video"],4))
PY
done
