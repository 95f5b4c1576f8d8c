cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
d=gpurun_out/prof_try_2x5_nographs; mkdir -p $d
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python bench.py --no-cpu-baseline --no-vae --no-detail --no-probe --no-graphs --steps 10 --warmup 10 > $d/bench.json 2> $d/bench.err; echo "trace 2x5 no-graphs rc=$?"
find $d -name "*kernel_trace.csv" -delete
grep '^{' $d/bench.json | cut -c1-200; tail -n 3 $d/bench.err | cut -c1-200
d=gpurun_out/prof_try_2x3_b; mkdir -p $d
HSA_ENABLE_DEBUG=0 timeout 900 rocprofv3 --kernel-trace --output-format csv -d $d -- python bench.py --no-cpu-baseline --no-vae --no-detail --no-probe --steps 6 --warmup 6 > $d/bench.json 2> $d/bench.err; echo "trace 2x3 graphs, no --stats rc=$?"
find $d -name "*kernel_trace.csv" -size +200M -delete; ls -la $d/*/ 2>/dev/null | head
