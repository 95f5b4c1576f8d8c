cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python tools/k320_ab.py --batch 5 > gpurun_out/r06_k320_ab.jsonl 2> gpurun_out/r06_k320_ab.err; cat gpurun_out/r06_k320_ab.jsonl; tail -n 3 gpurun_out/r06_k320_ab.err
timeout 600 python tools/k320_ab.py --batch 1 > gpurun_out/r06_k320_ab_v1.jsonl 2>> gpurun_out/r06_k320_ab.err; cat gpurun_out/r06_k320_ab_v1.jsonl
timeout 900 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-vae > gpurun_out/r06h_bench.log 2> gpurun_out/r06h_bench.err; echo "bench rc=$?"; grep '^{' gpurun_out/r06h_bench.log | tail -n 1 | cut -c1-3000; tail -n 3 gpurun_out/r06h_bench.err
