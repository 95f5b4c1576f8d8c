#!/bin/bash
# Round 5, GPU call 4: SQ counters of gemm5 vs the vendor kernel on the wide-N layers, package power / clock of the timed regime,
# packing of independent videos with the shared prefix, and the standing validation of the final tree (suite + driver's command).
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/pmc_sq.sh widen > gpurun_out/r05_pmc_sq_widen.md 2>&1; tail -n 12 gpurun_out/r05_pmc_sq_widen.md | cut -c1-400
bash tools/smi_power.sh r05_pw_3lanes --steps 9 --warmup 3 --no-cpu-baseline --no-vae --no-detail | tee gpurun_out/r05_power_clock.jsonl
for pk in "3 1" "4 1" "2 2" "3 2"; do
  set -- $pk
  timeout 500 python bench.py --steps 12 --warmup 3 --inflight $1 --batch $2 --no-cpu-baseline --no-vae --no-detail --no-probe > gpurun_out/r05_pack_$1x$2.log 2>&1
  echo "packing $1 lanes x $2 batched: $(grep '^{' gpurun_out/r05_pack_$1x$2.log | tail -n 1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["peak_reserved_gib"])')" | tee -a gpurun_out/r05_packing.txt
done
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=5 > gpurun_out/r05_pytest_gpu_final2.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/r05_pytest_gpu_final2.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_final2.log 2> gpurun_out/r05_bench_final2.err
echo "driver-like bench rc=$?"; grep '^{' gpurun_out/r05_bench_final2.log | tail -n 1 > gpurun_out/r05_bench_final2_line.json; cut -c1-1400 gpurun_out/r05_bench_final2_line.json; echo
