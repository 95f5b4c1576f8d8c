#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py tests/test_fullsize_parity.py -m gpu -q -p no:cacheprovider -k "outgrow or negative_control or outlier" -s > gpurun_out/r05_call5_pytest.log 2>&1
echo "pytest rc=$?"; tail -n 6 gpurun_out/r05_call5_pytest.log | cut -c1-300; grep "PARITY\|TRANS_HAZARD" gpurun_out/r05_call5_pytest.log | cut -c1-700
