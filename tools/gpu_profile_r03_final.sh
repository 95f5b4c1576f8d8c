#!/bin/bash
# final round-3 measurements: the driver's bench command, rocprofv3 kernel-trace stats of (a) the default command (hipGraph
# replay, three videos in flight) and (b) one video at a time on the eager launch sequence, the other BASELINE configs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r03_a gpurun_out/prof_r03_b
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_final.json 2> gpurun_out/r03_bench_final.err
echo "driver-like bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r03_bench_final.json')); print({k: d[k] for k in ('value','ms_per_step','e2e_frac_of_mfma_peak','roofline_coverage_of_probe_video')}, d['roofline']['kernel'], d['roofline']['frac'], d['eager']['videos_per_min'], d['hbm_footprint'], d.get('cpu_baseline',{}).get('value'), d.get('reference_gpu_baseline',{}).get('videos_per_min'))"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_a -- python bench.py --no-cpu-baseline --no-vae --steps 3 > gpurun_out/prof_r03_a/bench.json 2> gpurun_out/prof_r03_a/bench.err
echo "trace a rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_b -- python bench.py --no-cpu-baseline --no-vae --no-graphs --inflight 1 --steps 2 > gpurun_out/prof_r03_b/bench.json 2> gpurun_out/prof_r03_b/bench.err
echo "trace b rc=$?"
find gpurun_out/prof_r03_a gpurun_out/prof_r03_b -name "*kernel_trace.csv" -delete
python tools/kernel_stats_md.py gpurun_out/prof_r03_a gpurun_out/prof_r03_b gpurun_out/r03_kernel_stats.md "round-3"
cut -c1-200 gpurun_out/prof_r03_a/bench.json; echo; cut -c1-200 gpurun_out/prof_r03_b/bench.json; echo
timeout 600 python bench.py --no-cpu-baseline --no-vae --frames 16 --size 256 --ddim-steps 10 --guided-steps 5 --guidance-scale 0.3 --steps 16 --warmup 8 --inflight 8 > gpurun_out/r03_bench_cfg1.json 2> gpurun_out/r03_bench_cfg1.err
timeout 600 python bench.py --no-cpu-baseline --no-vae --sparsectrl --guided-steps 12 --guidance-scale 0.3 --steps 6 --warmup 3 > gpurun_out/r03_bench_cfg4.json 2> gpurun_out/r03_bench_cfg4.err
timeout 900 python bench.py --no-cpu-baseline --no-vae --frames 32 --size 768 --ddim-steps 50 --guided-steps 30 --steps 2 --warmup 2 --inflight 2 > gpurun_out/r03_bench_cfg5.json 2> gpurun_out/r03_bench_cfg5.err
for c in 1 4 5; do python -c "
import json; d=json.load(open('gpurun_out/r03_bench_cfg$c.json')); print('cfg$c', d['value'], d['config']['workload'][:60], d['hbm_footprint']['peak_reserved_gib'])" || tail -3 gpurun_out/r03_bench_cfg$c.err; done
