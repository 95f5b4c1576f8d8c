#!/bin/bash
# Round 5, last GPU call: the -m gpu suite and smoke() on the final tree
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=5 > gpurun_out/r05_pytest_gpu_final4.log 2>&1; echo "pytest rc=$?"; tail -n 4 gpurun_out/r05_pytest_gpu_final4.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -n 2
