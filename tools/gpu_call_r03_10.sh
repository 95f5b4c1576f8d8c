#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python tools/lanes_gpu.py --lanes 3 --videos 6 > gpurun_out/r03_lanes_gpu.json 2> gpurun_out/r03_lanes_gpu.err
echo "lanes rc=$?"; cat gpurun_out/r03_lanes_gpu.json; grep -v "it/s\|^$" gpurun_out/r03_lanes_gpu.err | head -30 | cut -c1-300
