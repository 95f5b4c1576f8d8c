#!/bin/bash
# Round 5, GPU call 2: vendor kernel names on the wide-N shapes; footprint A/B (tape closures dropped as they run; guided steps
# as two B = 1 forwards)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/vendor_names
R=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/vendor_names -- python $R/tools/vendor_kernel_names.py > $R/gpurun_out/vendor_names/run.log 2>&1 )
echo "vendor names rc=$?"
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/vendor_names/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
seen = set()
for r in rows:
    n = r["Kernel_Name"]
    if "Cijk" in n or "gemm" in n.lower() or "matmul" in n.lower():
        key = (n, r.get("Grid_Size"), r.get("Workgroup_Size"))
        if key in seen: continue
        seen.add(key)
        print({k: r.get(k) for k in ("Kernel_Name", "Grid_Size", "Grid_Size_X", "Workgroup_Size", "Workgroup_Size_X", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count") if r.get(k) is not None},
              "dur_us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
PY
for v in default twob1 default twob1; do
  f=""; [ $v = twob1 ] && f="--two-b1-guided"
  timeout 400 python bench.py --steps 9 --warmup 3 --no-cpu-baseline --no-vae --no-detail $f > gpurun_out/r05_bench_fp_$v.log 2>&1
  echo "footprint $v: $(grep '^{' gpurun_out/r05_bench_fp_$v.log | tail -n 1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["eager_one_video_at_a_time_videos_per_min"], d["identical_to_eager_path"], d["peak_reserved_gib"])')" | tee -a gpurun_out/r05_bench_footprint_ab.txt
done
tail -n 2 gpurun_out/r05_bench_fp_twob1.log | cut -c1-600
