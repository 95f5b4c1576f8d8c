cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_operand_fuzz.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^E  .*(Error|assert)|^FAILED|passed|failed" | cut -c1-300 > gpurun_out/r06d_fuzz.log; tail -n 25 gpurun_out/r06d_fuzz.log
for cfg in "2 5" "2 6" "1 8" "3 4" "2 5"; do
  set -- $cfg
  n=$(( $1 * $2 * 2 ))
  timeout 700 python bench.py --steps $n --warmup $(( $1 * $2 )) --inflight $1 --batch $2 --no-cpu-baseline --no-vae --no-detail --no-probe > gpurun_out/r06_pack_$1x$2.log 2>&1
  echo "lanes $1 x batch $2: $(grep '^{' gpurun_out/r06_pack_$1x$2.log | tail -n 1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["peak_reserved_gib"], d.get("cold_start_seconds"))')" | tee -a gpurun_out/r06_packing.txt
done
