#!/bin/bash
# The round's record in ONE GPU lease: the -m gpu suite, smoke, rocprofv3 kernel-trace stats of (a) the TIMED regime only (hipGraph
# replay, the driver's packing, no probe videos: `--no-probe`) -> profiles/kernel_durations_timed.json (bench.py's roofline_timed: it
# carries the library's source stamp, the device and the packing) and (b) one video at a time on the eager launch sequence, then the
# driver's bench command, PMC HBM-traffic passes per shape, the other BASELINE configs.  Everything lands under gpurun_out/
# (copied into profiles/ by hand).     gpurun --timeout 2700 -- 'bash tools/gpu_profile.sh r06 [skip-pytest]'
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
T=${1:-r06}
mkdir -p gpurun_out/prof_${T}_a gpurun_out/prof_${T}_b
if [ "$2" != "skip-pytest" ] && [ "$2" != "traces-only" ]; then
  timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > gpurun_out/${T}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 4 gpurun_out/${T}_pytest_gpu.log
  timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -n 2
fi
# (a) runs WITHOUT hipGraphs in round 6: rocprofv3 --kernel-trace (ROCm 7.2) segfaults in a tool thread ~20 s into the replay of graphs
# captured with >= 3 videos batched per lane (1 x 5, 2 x 3, 2 x 5; 3 x 1 and 2 x 2 trace fine; the un-profiled runs and the whole GPU
# suite are clean).  The eager run issues the same kernels in the same order on the same two streams with the same packing; only the
# launch mechanism differs (32.1 instead of 38 videos/min under the profiler: host launch gaps, which a per-kernel duration does not see)
export MC_PROFILE_CMD_A="python bench.py --no-cpu-baseline --no-vae --no-detail --no-probe --no-graphs --steps 10 --warmup 10"
# (b) = the regime of the bench's roofline probe: ONE lane's job (five videos batched into one launch sequence) alone on the eager launches
export MC_PROFILE_CMD_B="python bench.py --no-cpu-baseline --no-vae --no-detail --no-probe --no-graphs --inflight 1 --batch 5 --gemm-lanes 2 --steps 5 --warmup 5"
timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${T}_a -- $MC_PROFILE_CMD_A > gpurun_out/prof_${T}_a/bench.json 2> gpurun_out/prof_${T}_a/bench.err
echo "trace a rc=$?"
timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${T}_b -- $MC_PROFILE_CMD_B > gpurun_out/prof_${T}_b/bench.json 2> gpurun_out/prof_${T}_b/bench.err
echo "trace b rc=$?"
# videos in the traces: (a) 10 warm-up + 10 timed; (b) 5 warm-up + 5 timed
python tools/kernel_stats_md.py gpurun_out/prof_${T}_a gpurun_out/prof_${T}_b gpurun_out/${T}_kernel_stats.md "round-6" 20 10 || echo "kernel_stats_md failed"
find gpurun_out/prof_${T}_a gpurun_out/prof_${T}_b -name "*kernel_trace.csv" -delete
python -c "
import json; d=json.load(open('gpurun_out/kernel_durations_timed.json')); print('timed durations:', len(d['kernels']), 'kernels, total', round(d['total_kernel_s'],2), 's, overlap', d['overlap'], 'lanes x batch', d['lanes'], d['batch'], 'stamp', str(d['lib_stamp'])[:12], d['device'])"
cp gpurun_out/kernel_durations_timed.json profiles/kernel_durations_timed.json    # what the bench line's roofline_timed reads (same box, same code)
if [ "$2" = "traces-only" ]; then exit 0; fi
timeout 1000 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_final.log 2> gpurun_out/${T}_bench_final.err
echo "driver-like bench rc=$?"; grep '^{' gpurun_out/${T}_bench_final.log | tail -n 1 > gpurun_out/${T}_bench_final_line.json; cut -c1-2200 gpurun_out/${T}_bench_final_line.json; echo
cp gpurun_out/r06_bench_detail.json gpurun_out/${T}_bench_final_detail.json
# HBM traffic per shape in the tile choice of the timed region
mkdir -p gpurun_out/pmc_${T}
for c in FETCH_SIZE WRITE_SIZE; do
  PMC_LANES=2 PMC_BATCH=5 timeout 400 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${T}/$c -- python tools/pmc_traffic.py run gpurun_out/pmc_${T} > gpurun_out/pmc_${T}/$c.log 2>&1; echo "pmc $c rc=$?"
done
python tools/pmc_traffic.py table gpurun_out/pmc_${T} > gpurun_out/${T}_hbm_traffic_per_shape.json 2> gpurun_out/${T}_pmc_hbm_traffic.md || echo "pmc table failed"
tail -n 17 gpurun_out/${T}_pmc_hbm_traffic.md
timeout 600 python bench.py --no-cpu-baseline --no-vae --no-detail --frames 16 --size 256 --ddim-steps 10 --guided-steps 5 --guidance-scale 0.3 --steps 16 --warmup 8 --inflight 8 > gpurun_out/${T}_bench_cfg1.json 2> gpurun_out/${T}_bench_cfg1.err
timeout 600 python bench.py --no-cpu-baseline --no-vae --no-detail --sparsectrl --guided-steps 12 --guidance-scale 0.3 --steps 6 --warmup 3 > gpurun_out/${T}_bench_cfg4.json 2> gpurun_out/${T}_bench_cfg4.err
timeout 900 python bench.py --no-cpu-baseline --no-vae --no-detail --frames 32 --size 768 --ddim-steps 50 --guided-steps 30 --steps 2 --warmup 2 --inflight 2 > gpurun_out/${T}_bench_cfg5.json 2> gpurun_out/${T}_bench_cfg5.err
for c in 1 4 5; do python -c "
import json; d=json.loads([l for l in open('gpurun_out/${T}_bench_cfg$c.json') if l.startswith('{')][-1]); print('cfg$c', d['value'], d['config']['workload'][:60], d.get('peak_reserved_gib'))" || tail -n 3 gpurun_out/${T}_bench_cfg$c.err; done
