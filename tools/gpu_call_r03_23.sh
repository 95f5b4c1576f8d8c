#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -k "attention" 2>&1 | tail -2
timeout 900 python tools/lanes_gpu.py --lanes 3 --videos 6 > gpurun_out/r03_lanes_gpu_final.json 2> gpurun_out/r03_lanes_gpu_final.err
echo "lanes rc=$?"; cat gpurun_out/r03_lanes_gpu_final.json | cut -c1-1500; grep -v "it/s\|^$" gpurun_out/r03_lanes_gpu_final.err | tail -5 | cut -c1-200
