cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels.py tests/test_engine_modules.py -m gpu -q -p no:cacheprovider -k "groupnorm_statistics_of_its_output or norm_gemm or resnet or spatial or motion" 2>&1 | tail -n 5
for v in A B A B A B; do
  if [ $v = A ]; then X="--no-gn-epilogue"; else X=""; fi
  timeout 500 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-vae --no-detail --no-probe $X > gpurun_out/r06_gn_ab_$v.log 2>&1
  echo "$v ($X) $(grep '^{' gpurun_out/r06_gn_ab_$v.log | tail -n 1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')" | tee -a gpurun_out/r06_gn_epilogue_abab2.txt
done
