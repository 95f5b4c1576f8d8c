cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider -k "two_workgroups or gemm5_ring or norm_gemm" 2>&1 | tail -3
timeout 300 python tools/gemm5_bench.py --rounds 4 --iters 10 --variants 11,9 --only l > gpurun_out/r04_gemm5_2wg_micro.jsonl 2>gpurun_out/r04_gemm5_2wg_micro.err
python - <<'PY'
import json
for l in open("gpurun_out/r04_gemm5_2wg_micro.jsonl"):
    if l.startswith("{"):
        r=json.loads(l)
        if "v9_us" in r: print("%-28s M=%6d N=%5d K=%5d  8-wave %7.1f us  2wg %7.1f us  (%.0f -> %.0f TF)"%(r["shape"],r["M"],r["N"],r["K"],r["v11_us"],r["v9_us"],r["v11_TF"],r["v9_TF"]))
PY
export MC_HIP_LIB=$PWD/tools/_build/libmotionclone_hip_tools.so
for g in 0 1 2 3; do
MC_GEMM5_2WG=$g timeout 400 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-vae --no-detail --shapes-out gpurun_out/r04_shapes_2wg$g.json > gpurun_out/r04_bench_2wg$g.log 2>&1; echo "2WG=$g $(grep '^{' gpurun_out/r04_bench_2wg$g.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["eager_one_video_at_a_time_videos_per_min"], d["roofline"]["kernel"], d["roofline"]["frac"])')"
done
