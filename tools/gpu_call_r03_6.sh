#!/bin/bash
# round 3, GPU call 6: split-K A/B (argparse fix), e2e with power / clock sampling, single-lane A/B gemm3 vs gemm5
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python tools/gemm5_bench.py --variants=-1,0,11,15 --only small > gpurun_out/r03_gemm5_splitk.jsonl 2> gpurun_out/r03_gemm5_splitk.err
timeout 900 python tools/gemm5_bench.py --variants=-1,0,11,15 --only _l2 >> gpurun_out/r03_gemm5_splitk.jsonl 2>> gpurun_out/r03_gemm5_splitk.err
echo "bench rc=$?"; python - <<'PY'
import json
seen=set()
for ln in open("gpurun_out/r03_gemm5_splitk.jsonl"):
    if ln.startswith("{"):
        r = json.loads(ln)
        if r["shape"] in seen: continue
        seen.add(r["shape"])
        print("%-30s M=%-6d N=%-5d K=%-5d | " % (r["shape"], r["M"], r["N"], r["K"]) + " | ".join("%s %6.1f us %4.0f TF" % (n, r.get("v%d_us" % v, 0), r.get("v%d_TF" % v, 0)) for n, v in (("auto-r2-splitk", -1), ("auto", 0), ("g5-256", 11), ("g5-128", 15))))
PY
tail -3 gpurun_out/r03_gemm5_splitk.err
rocm-smi --showpower --showclocks 2>/dev/null | head -30
(while true; do rocm-smi --showpower --showclocks --csv 2>/dev/null | tail -n +2 | head -2 | tr '\n' ' '; echo; sleep 0.5; done) > gpurun_out/r03_smi_3lane.log 2>&1 &
SMI=$!
timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-vae > gpurun_out/r03_bench_3lane.json 2> gpurun_out/r03_bench_3lane.err
kill $SMI
echo "3 lanes:"; python -c "
import json; d=json.load(open('gpurun_out/r03_bench_3lane.json')); print(d['value'], d.get('eager',{}).get('videos_per_min'))"
tail -25 gpurun_out/r03_smi_3lane.log | cut -c1-300
(while true; do rocm-smi --showpower --showclocks --csv 2>/dev/null | tail -n +2 | head -2 | tr '\n' ' '; echo; sleep 0.5; done) > gpurun_out/r03_smi_1lane.log 2>&1 &
SMI=$!
timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-vae --inflight 1 > gpurun_out/r03_bench_1lane_g5.json 2> gpurun_out/r03_bench_1lane_g5.err
kill $SMI
echo "1 lane gemm5:"; python -c "
import json; d=json.load(open('gpurun_out/r03_bench_1lane_g5.json')); print(d['value'], d.get('eager',{}).get('videos_per_min'))"
tail -12 gpurun_out/r03_smi_1lane.log | cut -c1-300
MC_NO_GEMM5=1 timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-vae --inflight 1 > gpurun_out/r03_bench_1lane_g3.json 2> gpurun_out/r03_bench_1lane_g3.err
echo "1 lane gemm3:"; python -c "
import json; d=json.load(open('gpurun_out/r03_bench_1lane_g3.json')); print(d['value'], d.get('eager',{}).get('videos_per_min'))"
