#!/bin/bash
# Round 5, GPU call 6: the standing validation of the tree with the attention fix (suite + smoke + the driver's command), and the
# short A/B the fix deserves (it adds one s_nop in a rarely taken branch: no change expected)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=5 > gpurun_out/r05_pytest_gpu_final3.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/r05_pytest_gpu_final3.log
grep "TRANS_HAZARD" gpurun_out/r05_pytest_gpu_final3.log | cut -c1-400
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -n 2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_final3.log 2> gpurun_out/r05_bench_final3.err
echo "driver-like bench rc=$?"; grep '^{' gpurun_out/r05_bench_final3.log | tail -n 1 > gpurun_out/r05_bench_final3_line.json; cut -c1-1600 gpurun_out/r05_bench_final3_line.json; echo
cp gpurun_out/r05_bench_detail.json gpurun_out/r05_bench_final3_detail.json
