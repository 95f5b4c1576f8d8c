#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fullsize_parity.py -m gpu -q -p no:cacheprovider -k "other_frame_counts" -s > gpurun_out/r05_call10_pytest.log 2>&1
echo "pytest rc=$?"; tail -n 25 gpurun_out/r05_call10_pytest.log | cut -c1-300; grep "PARITY" gpurun_out/r05_call10_pytest.log | cut -c1-700
