#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider -k "outgrow or negative_control" > gpurun_out/r05_call8_pytest.log 2>&1
echo "pytest rc=$?"; tail -n 12 gpurun_out/r05_call8_pytest.log | cut -c1-300
