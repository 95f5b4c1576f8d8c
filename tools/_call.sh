cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_kernels.py tests/test_engine_modules.py -m gpu -q -p no:cacheprovider -k "norm_gemm or level0" 2>&1 | tail -4
timeout 300 python tools/norm_gemm_bench.py > gpurun_out/r04_norm_gemm_v2.jsonl 2> gpurun_out/r04_norm_gemm.err; cat gpurun_out/r04_norm_gemm_v2.jsonl; tail -2 gpurun_out/r04_norm_gemm.err
timeout 400 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-vae --shapes-out gpurun_out/r04_shapes_fused2.json > gpurun_out/r04_bench_fused2.log 2>&1; grep '^{' gpurun_out/r04_bench_fused2.log | cut -c1-300; cp gpurun_out/r04_bench_detail.json gpurun_out/r04_bench_fused2_detail.json
