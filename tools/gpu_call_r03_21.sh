#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels.py tests/test_engine_modules.py tests/test_engine_parity.py tests/test_vae.py -m gpu -q -k "groupnorm or resnet or vae or transformer or motion or parity or cross" 2>&1 | tail -4
timeout 900 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-vae --shapes-out gpurun_out/r03_shapes_in_situ2.json > gpurun_out/r03_bench_gnfold.json 2>gpurun_out/r03_bench_gnfold.err
python - <<PY
import json
l=[x for x in open("gpurun_out/r03_bench_gnfold.json") if x.startswith("{")]
j=json.loads(l[-1]); print("videos/min", j["value"], "ms/video", j["ms_per_step"], "e2e frac", j["e2e_frac_of_mfma_peak"])
print("launches", sum(v["launches"] for v in j["roofline_by_kernel"].values()))
for k,v in sorted(j["roofline_by_kernel"].items(), key=lambda kv:-kv[1]["share_of_probe_video"])[:14]: print("  %-44s n=%5d avg=%7.1f frac=%.3f share=%.4f"%(k,v["launches"],v["avg_launch_us"],v["frac"],v["share_of_probe_video"]))
PY
