#!/bin/bash
# round 3, GPU call 7: new bench probe (every C-ABI launch), PMC HBM traffic per shape, GPU test suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_traffic
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_traffic/$c -- python tools/pmc_traffic.py run gpurun_out/pmc_traffic > gpurun_out/pmc_traffic/$c.log 2>&1
  echo "pmc $c rc=$?"
done
python tools/pmc_traffic.py table gpurun_out/pmc_traffic > gpurun_out/hbm_traffic_per_shape.json 2> gpurun_out/hbm_traffic_per_shape.md
echo "table rc=$?"; cat gpurun_out/hbm_traffic_per_shape.md
mkdir -p profiles; cp gpurun_out/hbm_traffic_per_shape.json profiles/hbm_traffic_per_shape.json
timeout 900 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-vae --shapes-out gpurun_out/r03_shapes.json > gpurun_out/r03_bench_probe.json 2> gpurun_out/r03_bench_probe.err
echo "bench rc=$?"; tail -3 gpurun_out/r03_bench_probe.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_bench_probe.json"))
print("value", d["value"], "eager", d["eager"], "coverage", d["roofline_coverage_of_probe_video"], "hbm", d["hbm_footprint"])
for k, v in sorted(d["roofline_by_kernel"].items(), key=lambda kv: -kv[1]["share_of_probe_video"]):
    print("%-44s %-4s frac %.3f (mfma %.3f hbm %.3f) launches %5d avg %7.1f us share %.3f traffic %s" % (
        k, v["bound"], v["frac"], v["frac_of_mfma_peak"], v["frac_of_hbm_peak"], v["launches"], v["avg_launch_us"],
        v["share_of_probe_video"], v.get("traffic_vs_algorithmic")))
PY
