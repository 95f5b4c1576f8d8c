#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/r03_attn_qt_ab.jsonl
for q in 0 2; do MC_ATTN_QT=$q MC_ATTN_TAG="ring,qt=$q" timeout 300 python tools/attn_bench.py --fwd-only 2>/dev/null | grep '"l0"' >> gpurun_out/r03_attn_qt_ab.jsonl; done
cat gpurun_out/r03_attn_qt_ab.jsonl
MC_ATTN_QT=2 timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -k "ring_kernel or long_sequence" 2>&1 | tail -2
