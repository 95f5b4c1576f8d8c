#!/bin/bash
# round-2 GPU call 1: full-size parity (oracle fp32 on the GPU), reference-on-GPU timing, bench baseline of this box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== fullsize parity" 
timeout 1500 python -m pytest tests/test_fullsize_parity.py -q -m gpu -s -x > gpurun_out/fullsize_parity.log 2>&1
echo "rc=$?"; tail -15 gpurun_out/fullsize_parity.log
echo "== reference on GPU (stock PyTorch-ROCm)"
for args in "--size 256 --miopen 0 --dtype f16" "--size 256 --miopen 1 --dtype f16" "--size 512 --miopen 0 --dtype f16" "--size 256 --miopen 0 --dtype f32"; do
  timeout 600 python tools/ref_gpu_timing.py $args 2>gpurun_out/ref_gpu_err.log | tail -1 | tee -a gpurun_out/ref_gpu_timing.jsonl
done
echo "== bench"
timeout 600 python bench.py --steps 2 --warmup 1 --no-vae > gpurun_out/bench_call1.json 2>gpurun_out/bench_call1.err
echo "rc=$?"; cut -c1-600 gpurun_out/bench_call1.json
