cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_operand_fuzz.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^E  .*(Error|assert)|^FAILED|passed|failed" | cut -c1-300 > gpurun_out/r06f_fuzz.log; tail -n 8 gpurun_out/r06f_fuzz.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06g_bench_stdout.log 2> gpurun_out/r06g_bench_stderr.log
echo "bench rc=$?"; grep '^{' gpurun_out/r06g_bench_stdout.log | tail -n 1 | tee gpurun_out/r06g_bench_line.json | cut -c1-2500; tail -n 5 gpurun_out/r06g_bench_stderr.log
cp profiles/r06_bench_detail.json gpurun_out/r06g_bench_detail.json 2>/dev/null
