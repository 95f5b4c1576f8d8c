#!/bin/bash
# round 3, GPU call 5: split-K on gemm5 (native slabs + reduce5) vs round-2 split-K (gemm3 + splitk_reduce); end-to-end
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "gemm" > gpurun_out/r03_pytest_gemm5c.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/r03_pytest_gemm5c.log
timeout 900 python tools/gemm5_bench.py --variants -1,0,11,15 --only small > gpurun_out/r03_gemm5_splitk.jsonl 2> gpurun_out/r03_gemm5_splitk.err
timeout 900 python tools/gemm5_bench.py --variants -1,0,11,15 --only _l2 >> gpurun_out/r03_gemm5_splitk.jsonl 2>> gpurun_out/r03_gemm5_splitk.err
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
timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-vae > gpurun_out/r03_bench_call5.json 2> gpurun_out/r03_bench_call5.err
echo "bench rc=$?"; cut -c1-300 gpurun_out/r03_bench_call5.json; python -c "
import json; d=json.load(open('gpurun_out/r03_bench_call5.json')); print(d['value'], d.get('eager',{}).get('videos_per_min'), d.get('sec_per_guided_step'), d.get('sec_per_plain_step'))"
