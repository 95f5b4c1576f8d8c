cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "2 2" "1 2" "1 3" "2 3" "1 4"; do set -- $cfg
timeout 500 python bench.py --steps 12 --warmup 3 --inflight $1 --batch $2 --no-cpu-baseline --no-vae --no-detail > gpurun_out/r04_bench_if$1_b$2.log 2>&1; echo "inflight=$1 batch=$2 $(grep '^{' gpurun_out/r04_bench_if$1_b$2.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["eager_one_video_at_a_time_videos_per_min"], d["identical_to_eager_path"], d["peak_reserved_gib"])' 2>&1 | tail -1)"; tail -2 gpurun_out/r04_bench_if$1_b$2.log | grep -i "error\|Traceback" 
done
