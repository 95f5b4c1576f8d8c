#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_logit_range_stress.py -m gpu -q -p no:cacheprovider > gpurun_out/r05_call6_pytest.log 2>&1
echo "pytest rc=$?"; tail -n 40 gpurun_out/r05_call6_pytest.log | cut -c1-400
