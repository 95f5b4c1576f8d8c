#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels.py tests/test_edge_cases.py tests/test_engine_modules.py tests/test_determinism.py -m gpu -q -k "temporal or motion or attention or determin or bit" 2>&1 | tail -4
: > gpurun_out/r03_tattn_vec_ab.jsonl
for x in 0 1; do MC_TATTN_VEC=$x timeout 300 python tools/tattn_bench.py >> gpurun_out/r03_tattn_vec_ab.jsonl 2>/dev/null; done
cat gpurun_out/r03_tattn_vec_ab.jsonl
