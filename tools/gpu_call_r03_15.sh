#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/r03_pytest_gpu_call15.log
timeout 900 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-vae --shapes-out gpurun_out/r03_shapes_in_situ.json > gpurun_out/r03_bench_attn_ring2.json 2>gpurun_out/r03_bench_attn_ring2.err
python - <<PY
import json
l=[x for x in open("gpurun_out/r03_bench_attn_ring2.json") if x.startswith("{")]
j=json.loads(l[-1]); print("videos/min", j["value"], "ms/video", j["ms_per_step"], "e2e frac", j["e2e_frac_of_mfma_peak"])
for k,v in j.get("roofline_by_kernel",{}).items():
    if k.startswith("attn"): print("  ",k, round(v["avg_launch_us"],1), round(v["frac"],3), round(v["share_of_probe_video"],4))
PY
