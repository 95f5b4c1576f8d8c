#!/bin/bash
# round-2 profiling call: rocprofv3 kernel trace of the bench (eager launch sequence) + PMC HBM traffic of the probe shapes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r02 gpurun_out/pmc_r02
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02 -- python bench.py --no-cpu-baseline --no-vae --no-graphs --steps 2 --warmup 1 > gpurun_out/prof_r02/bench.json 2> gpurun_out/prof_r02/bench.err
echo "trace rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_r02/$c -- python tools/pmc_probe.py > gpurun_out/pmc_r02/$c.log 2>&1
  echo "pmc $c rc=$?"
done
find gpurun_out/prof_r02 gpurun_out/pmc_r02 -name "*.csv" | head -20
# keep only the small summaries (the raw kernel trace is large)
find gpurun_out/prof_r02 -name "*kernel_trace.csv" -size +20M -delete
ls -la gpurun_out/prof_r02/*/* gpurun_out/pmc_r02/*/*/* 2>/dev/null | head -30
