#!/bin/bash
# Round-5 record: the -m gpu suite, smoke, the driver's bench command, rocprofv3 kernel-trace stats of (a) the TIMED regime only
# (hipGraph replay, three videos in flight, no probe videos: `--no-probe`) -> profiles/kernel_durations_timed.json (bench.py's
# roofline_timed) and (b) one video at a time on the eager launch sequence, PMC HBM-traffic passes per shape in the three-lane tile
# choice, the other BASELINE configs.  Everything under gpurun_out/ (copied into profiles/ by hand).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
T=${1:-r05}
mkdir -p gpurun_out/prof_${T}_a gpurun_out/prof_${T}_b
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > gpurun_out/${T}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 4 gpurun_out/${T}_pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -n 2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${T}_a -- python bench.py --no-cpu-baseline --no-vae --no-detail --no-probe --steps 6 --warmup 3 > gpurun_out/prof_${T}_a/bench.json 2> gpurun_out/prof_${T}_a/bench.err
echo "trace a rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${T}_b -- python bench.py --no-cpu-baseline --no-vae --no-detail --no-graphs --inflight 1 --steps 2 > gpurun_out/prof_${T}_b/bench.json 2> gpurun_out/prof_${T}_b/bench.err
echo "trace b rc=$?"
python tools/kernel_stats_md.py gpurun_out/prof_${T}_a gpurun_out/prof_${T}_b gpurun_out/${T}_kernel_stats.md "round-5" 9 6 || echo "kernel_stats_md failed"
find gpurun_out/prof_${T}_a gpurun_out/prof_${T}_b -name "*kernel_trace.csv" -delete
python -c "
import json; d=json.load(open('gpurun_out/kernel_durations_timed.json')); print('timed durations:', len(d['kernels']), 'kernels, total', round(d['total_kernel_s'],2), 's, overlap', d['overlap'])"
cp gpurun_out/kernel_durations_timed.json profiles/kernel_durations_timed.json    # what the bench line's roofline_timed reads (same box, same code)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_final.log 2> gpurun_out/${T}_bench_final.err
echo "driver-like bench rc=$?"; grep '^{' gpurun_out/${T}_bench_final.log | tail -n 1 > gpurun_out/${T}_bench_final_line.json; cut -c1-1800 gpurun_out/${T}_bench_final_line.json; echo
cp gpurun_out/r05_bench_detail.json gpurun_out/${T}_bench_final_detail.json
# HBM traffic per shape in the tile choice of the timed region (three videos in flight)
mkdir -p gpurun_out/pmc_${T}_l3
for c in FETCH_SIZE WRITE_SIZE; do
  PMC_LANES=3 timeout 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${T}_l3/$c -- python tools/pmc_traffic.py run gpurun_out/pmc_${T}_l3 > gpurun_out/pmc_${T}_l3/$c.log 2>&1; echo "pmc lanes=3 $c rc=$?"
done
python tools/pmc_traffic.py table gpurun_out/pmc_${T}_l3 > gpurun_out/${T}_hbm_traffic_per_shape_l3.json 2> gpurun_out/${T}_pmc_hbm_traffic_l3.md || echo "pmc table failed"
tail -n 17 gpurun_out/${T}_pmc_hbm_traffic_l3.md
timeout 600 python bench.py --no-cpu-baseline --no-vae --no-detail --frames 16 --size 256 --ddim-steps 10 --guided-steps 5 --guidance-scale 0.3 --steps 16 --warmup 8 --inflight 8 > gpurun_out/${T}_bench_cfg1.json 2> gpurun_out/${T}_bench_cfg1.err
timeout 600 python bench.py --no-cpu-baseline --no-vae --no-detail --sparsectrl --guided-steps 12 --guidance-scale 0.3 --steps 6 --warmup 3 > gpurun_out/${T}_bench_cfg4.json 2> gpurun_out/${T}_bench_cfg4.err
timeout 900 python bench.py --no-cpu-baseline --no-vae --no-detail --frames 32 --size 768 --ddim-steps 50 --guided-steps 30 --steps 2 --warmup 2 --inflight 2 > gpurun_out/${T}_bench_cfg5.json 2> gpurun_out/${T}_bench_cfg5.err
for c in 1 4 5; do python -c "
import json; d=json.loads([l for l in open('gpurun_out/${T}_bench_cfg$c.json') if l.startswith('{')][-1]); print('cfg$c', d['value'], d['config']['workload'][:60], d.get('peak_reserved_gib'))" || tail -n 3 gpurun_out/${T}_bench_cfg$c.err; done
