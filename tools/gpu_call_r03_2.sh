#!/bin/bash
# round 3, GPU call 2: gemm5 (ring + stagger + wave-private epilogue) correctness on hardware, A/B against gemm3, end-to-end effect
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q -k "gemm" > gpurun_out/r03_pytest_gemm.log 2>&1
echo "pytest gemm rc=$?"; tail -3 gpurun_out/r03_pytest_gemm.log
timeout 900 python tools/gemm5_bench.py > gpurun_out/r03_gemm5_bench.jsonl 2> gpurun_out/r03_gemm5_bench.err
echo "bench rc=$?"; cat gpurun_out/r03_gemm5_bench.jsonl; tail -3 gpurun_out/r03_gemm5_bench.err
timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-vae > gpurun_out/r03_bench_g5.json 2> gpurun_out/r03_bench_g5.err
echo "bench g5 rc=$?"; cut -c1-400 gpurun_out/r03_bench_g5.json
MC_NO_GEMM5=1 timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-vae > gpurun_out/r03_bench_g3.json 2> gpurun_out/r03_bench_g3.err
echo "bench g3 rc=$?"; cut -c1-400 gpurun_out/r03_bench_g3.json
