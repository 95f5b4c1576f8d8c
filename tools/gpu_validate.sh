#!/bin/bash
# One GPU lease = the round's standing validation: the -m gpu suite, the driver's bench command, and (optionally) the
# rocprofv3 kernel trace of a short bench run.  Everything lands under gpurun_out/ (copied into profiles/ by hand).
#   gpurun --timeout 1500 -- 'bash tools/gpu_validate.sh [tag] [pytest|bench|prof ...]'
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
TAG=${1:-r05}; shift
WHAT=${*:-pytest bench}
mkdir -p gpurun_out
for w in $WHAT; do
  case $w in
    pytest)
      timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > gpurun_out/${TAG}_pytest_gpu.log 2>&1
      echo "pytest rc=$?"; tail -n 25 gpurun_out/${TAG}_pytest_gpu.log ;;
    bench)
      timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_stdout.log 2> gpurun_out/${TAG}_bench_stderr.log
      echo "bench rc=$?"; grep '^{' gpurun_out/${TAG}_bench_stdout.log | tail -n 1 | tee gpurun_out/${TAG}_bench_line.json | cut -c1-1500
      cp gpurun_out/r06_bench_detail.json gpurun_out/${TAG}_bench_detail.json 2>/dev/null ;;
    benchshort)
      timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-vae > gpurun_out/${TAG}_benchshort_stdout.log 2> gpurun_out/${TAG}_benchshort_stderr.log
      echo "benchshort rc=$?"; grep '^{' gpurun_out/${TAG}_benchshort_stdout.log | tail -n 1 | tee gpurun_out/${TAG}_benchshort_line.json | cut -c1-1200
      cp gpurun_out/r06_bench_detail.json gpurun_out/${TAG}_benchshort_detail.json 2>/dev/null ;;
    prof)
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/${TAG}_prof" -o trace -- \
          python "$OLDPWD/bench.py" --no-cpu-baseline --no-vae --steps 3 --no-detail > "$OLDPWD/gpurun_out/${TAG}_prof_bench.log" 2>&1 )
      echo "prof rc=$?"; find gpurun_out/${TAG}_prof -name '*kernel_stats.csv' | head -n 2
      # keep only the stats tables (the raw trace is hundreds of MB)
      find gpurun_out/${TAG}_prof -type f ! -name '*stats*.csv' -delete 2>/dev/null ;;
  esac
done
