#!/bin/bash
# round 3, GPU call 9: full GPU test suite, launcher lanes on hardware, videos-in-flight sweep
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest_gpu_2.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r03_pytest_gpu_2.log
timeout 900 python tools/lanes_gpu.py --lanes 3 --videos 6 > gpurun_out/r03_lanes_gpu.json 2> gpurun_out/r03_lanes_gpu.err
echo "lanes rc=$?"; cat gpurun_out/r03_lanes_gpu.json; tail -5 gpurun_out/r03_lanes_gpu.err
for nf in 2 4; do
  timeout 600 python bench.py --steps $((nf*2)) --warmup $nf --no-cpu-baseline --no-vae --inflight $nf > gpurun_out/r03_bench_if$nf.json 2> gpurun_out/r03_bench_if$nf.err
  python -c "
import json; d=json.load(open('gpurun_out/r03_bench_if$nf.json')); print('inflight $nf:', d['value'], 'hbm', d['hbm_footprint']['peak_reserved_gib'])"
done
