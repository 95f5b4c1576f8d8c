#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/gn_bench.py 2>/dev/null | tee gpurun_out/r03_gn_fused.jsonl
timeout 1200 python -m pytest tests/test_engine_parity.py tests/test_engine_modules.py tests/test_fullsize_parity.py tests/test_dropin_api.py -m gpu -q 2>&1 | tail -3
timeout 900 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-vae > gpurun_out/r03_bench_geglu_split.json 2>gpurun_out/r03_bench_geglu_split.err
python - <<PY
import json
l=[x for x in open("gpurun_out/r03_bench_geglu_split.json") if x.startswith("{")]
j=json.loads(l[-1]); print("videos/min", j["value"], "ms/video", j["ms_per_step"], "e2e frac", j["e2e_frac_of_mfma_peak"])
for k,v in j["roofline_by_kernel"].items():
    if "geglu" in k or "groupnorm" in k: print("  %-44s n=%5d avg=%7.1f frac=%.3f share=%.4f"%(k,v["launches"],v["avg_launch_us"],v["frac"],v["share_of_probe_video"]))
PY
