#!/bin/bash
# round 3, GPU call 4: gemm5 with 128-row tiles on the small levels vs gemm3's 128x320 (4 waves) / split-K choice
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "gemm5" > gpurun_out/r03_pytest_gemm5b.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/r03_pytest_gemm5b.log
timeout 900 python tools/gemm5_bench.py --variants 0,1,4,11,15 --only small > gpurun_out/r03_gemm5_small.jsonl 2> gpurun_out/r03_gemm5_small.err
timeout 900 python tools/gemm5_bench.py --variants 0,1,4,11,15 --only _l2 >> gpurun_out/r03_gemm5_small.jsonl 2>> gpurun_out/r03_gemm5_small.err
echo "bench rc=$?"; python - <<'PY'
import json
seen=set()
for ln in open("gpurun_out/r03_gemm5_small.jsonl"):
    if ln.startswith("{"):
        r = json.loads(ln)
        if r["shape"] in seen: continue
        seen.add(r["shape"])
        print("%-30s M=%-6d N=%-5d K=%-5d | " % (r["shape"], r["M"], r["N"], r["K"]) + " | ".join("%s %6.1f us %4.0f TF" % (n, r.get("v%d_us" % v, 0), r.get("v%d_TF" % v, 0)) for n, v in (("auto", 0), ("g3-256", 1), ("g3-128", 4), ("g5-256", 11), ("g5-128", 15))))
PY
tail -3 gpurun_out/r03_gemm5_small.err
