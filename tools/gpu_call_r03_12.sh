#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== tests, default (zero-extended remainder)"; timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -k "attention" 2>&1 | tail -8
echo "== tests, legacy remainder step"; MC_ATTN_LEG=1 timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -k "attention" 2>&1 | tail -8
: > gpurun_out/r03_attn_ring_ab.jsonl
for cfg in "0 0" "1 0" "1 1"; do set -- $cfg; MC_ATTN_RING=$1 MC_ATTN_LEG=$2 MC_ATTN_TAG="ring=$1,leg=$2" timeout 300 python tools/attn_bench.py --fwd-only >> gpurun_out/r03_attn_ring_ab.jsonl 2>gpurun_out/r03_attn_ring_ab.err; done
cat gpurun_out/r03_attn_ring_ab.jsonl; tail -3 gpurun_out/r03_attn_ring_ab.err
