#!/bin/bash
# final round-2 profile: rocprofv3 kernel trace + stats of (a) the default bench command (hipGraph replay, 3 videos in flight)
# and (b) the same workload one video at a time on the eager launch sequence (what the roofline probe pass of bench.py times)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_final_a gpurun_out/prof_final_b
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final_a -- python bench.py --no-cpu-baseline --no-vae --steps 3 > gpurun_out/prof_final_a/bench.json 2> gpurun_out/prof_final_a/bench.err
echo "trace a rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final_b -- python bench.py --no-cpu-baseline --no-vae --no-graphs --inflight 1 --steps 2 > gpurun_out/prof_final_b/bench.json 2> gpurun_out/prof_final_b/bench.err
echo "trace b rc=$?"
find gpurun_out/prof_final_a gpurun_out/prof_final_b -name "*kernel_trace.csv" -delete
find gpurun_out/prof_final_a gpurun_out/prof_final_b -name "*.csv" | head
cut -c1-200 gpurun_out/prof_final_a/bench.json; echo; cut -c1-200 gpurun_out/prof_final_b/bench.json
