#!/bin/bash
# round 3, GPU call 3: true A/B gemm3 vs gemm5 (explicit cfg = 1 is gemm3 again), SQ counters of both + attention
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/gemm5_bench.py --variants 1,11,12 > gpurun_out/r03_gemm5_bench2.jsonl 2> gpurun_out/r03_gemm5_bench2.err
echo "bench rc=$?"; python - <<'PY'
import json
for ln in open("gpurun_out/r03_gemm5_bench2.jsonl"):
    if ln.startswith("{"):
        r = json.loads(ln)
        print("%-26s gemm3 %7.1f us %5.0f TF | gemm5 %7.1f us %5.0f TF | gemm5-nostagger %7.1f us %5.0f TF | maxdiff %s" % (
            r["shape"], r.get("v1_us", 0), r.get("v1_TF", 0), r.get("v11_us", 0), r.get("v11_TF", 0), r.get("v12_us", 0), r.get("v12_TF", 0), r.get("v11_maxdiff_vs_gemm3")))
PY
bash tools/pmc_sq.sh gemm > gpurun_out/r03_pmc_sq_gemm.md 2>&1; cat gpurun_out/r03_pmc_sq_gemm.md
bash tools/pmc_sq.sh attn > gpurun_out/r03_pmc_sq_attn.md 2>&1; cat gpurun_out/r03_pmc_sq_attn.md
MC_GEMM5_VAR=1 timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-vae > gpurun_out/r03_bench_g5v1.json 2> gpurun_out/r03_bench_g5v1.err
echo "bench g5 var1 rc=$?"; cut -c1-300 gpurun_out/r03_bench_g5v1.json
