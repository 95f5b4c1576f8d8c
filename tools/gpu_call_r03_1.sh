#!/bin/bash
# round 3, GPU call 1: the parity holes (F = 32, config-2 trajectory, tightened tolerances), determinism stress + localisation
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -s > gpurun_out/r03_pytest_gpu_1.log 2>&1
echo "pytest rc=$?"
tail -5 gpurun_out/r03_pytest_gpu_1.log
grep -h "PARITY\|DETERMINISM\|top1 bit-exact" gpurun_out/r03_pytest_gpu_1.log | cut -c1-400
timeout 600 python tools/tattn_race.py run --runs 100 > gpurun_out/r03_tattn_race.jsonl 2> gpurun_out/r03_tattn_race.err
echo "race rc=$?"
cat gpurun_out/r03_tattn_race.jsonl
