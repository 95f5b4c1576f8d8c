cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools/pmc_sq.sh widen6 > gpurun_out/r06_pmc_sq_widen6.md 2> gpurun_out/r06_pmc_sq_widen6.err; tail -n 12 gpurun_out/r06_pmc_sq_widen6.md | cut -c1-700
( time timeout 1200 python bench.py > gpurun_out/r06i_bench.log 2> gpurun_out/r06i_bench.err ) 2>&1 | tail -n 4; echo "default bench rc=$?"; grep '^{' gpurun_out/r06i_bench.log | tail -n 1 | cut -c1-3200
