#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -k "attention" 2>&1 | tail -8
: > gpurun_out/r03_attn_l1_ring_ab.jsonl
for r in 0 1; do MC_ATTN_RING=$r MC_ATTN_TAG="ring=$r" timeout 300 python tools/attn_bench.py >> gpurun_out/r03_attn_l1_ring_ab.jsonl 2>/dev/null; done
grep '"l1"' gpurun_out/r03_attn_l1_ring_ab.jsonl
