#!/bin/bash
# round 3, GPU call 8: super-tile rasterisation of gemm5 - A/B on the wide layers, PMC traffic, end to end
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_traffic
timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "gemm" > gpurun_out/r03_pytest_gemm5d.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/r03_pytest_gemm5d.log
for v in 0 1; do
  MC_GEMM5_NO_SUPERTILE=$v timeout 600 python tools/gemm5_bench.py --variants=0,11 --only "l" > gpurun_out/r03_gemm5_swz$v.jsonl 2> gpurun_out/r03_gemm5_swz$v.err
done
python - <<'PY'
import json
rows = {}
for v in (0, 1):
    for ln in open("gpurun_out/r03_gemm5_swz%d.jsonl" % v):
        if ln.startswith("{"):
            r = json.loads(ln)
            rows.setdefault(r["shape"], {})[v] = r
for k, d in rows.items():
    if 0 in d and 1 in d:
        print("%-30s super-tile: auto %7.1f us %5.0f TF g5 %7.1f us | old order: auto %7.1f us g5 %7.1f us" % (
            k, d[0].get("v0_us", 0), d[0].get("v0_TF", 0), d[0].get("v11_us", 0), d[1].get("v0_us", 0), d[1].get("v11_us", 0)))
PY
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_traffic/$c -- python tools/pmc_traffic.py run gpurun_out/pmc_traffic > gpurun_out/pmc_traffic/$c.log 2>&1
  echo "pmc $c rc=$?"
done
python tools/pmc_traffic.py table gpurun_out/pmc_traffic > gpurun_out/hbm_traffic_per_shape.json 2> gpurun_out/hbm_traffic_per_shape.md
echo "table rc=$?"; cat gpurun_out/hbm_traffic_per_shape.md
timeout 900 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-vae > gpurun_out/r03_bench_swz.json 2> gpurun_out/r03_bench_swz.err
echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r03_bench_swz.json')); print('value', d['value'], 'eager', d['eager']['videos_per_min'])"
MC_GEMM5_NO_SUPERTILE=1 timeout 900 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-vae > gpurun_out/r03_bench_noswz.json 2> gpurun_out/r03_bench_noswz.err
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_noswz.json')); print('old order: value', d['value'], 'eager', d['eager']['videos_per_min'])"
