cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
export MC_HIP_LIB=$PWD/tools/_build/libmotionclone_hip_tools.so
for g in 0 4 0 4; do
MC_GEMM5_2WG=$g timeout 400 python bench.py --steps 9 --warmup 3 --no-cpu-baseline --no-vae --no-detail > gpurun_out/r04_bench_2wgsel$g.log 2>&1; echo "2WG=$g $(grep '^{' gpurun_out/r04_bench_2wgsel$g.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["eager_one_video_at_a_time_videos_per_min"], d["roofline"]["kernel"], d["roofline"]["frac"])')"
done
