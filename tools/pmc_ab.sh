# A/B of the conv tile order (MC_HIP_LIB = a build of the previous order): HBM bytes per level-0 conv launch from separate --pmc passes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/pmc_ab
for lib in before new; do
  for c in FETCH_SIZE WRITE_SIZE; do
    if [ $lib = before ]; then export MC_HIP_LIB=motionclone_amd/csrc/libmc_before.so; else unset MC_HIP_LIB; fi
    rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_ab/$lib/$c -- python tools/pmc_probe.py > gpurun_out/pmc_ab/$lib.$c.log 2>&1
    echo "$lib $c rc=$?"
  done
done
python - <<'PY'
import csv, glob
for lib in ("before", "new"):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("gpurun_out/pmc_ab/%s/%s/*/*counter_collection.csv" % (lib, c))
        if not f: print(lib, c, "no file"); continue
        for r in csv.DictReader(open(f[0])):
            if "gemm3_kernel<1, 256" in r["Kernel_Name"]:
                tot.setdefault(c, []).append(float(r["Counter_Value"]))
    for c, v in tot.items():
        print(lib, c, "launches", len(v), "mean KB", sum(v) / len(v))
    if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
        f = sum(tot["FETCH_SIZE"]) / len(tot["FETCH_SIZE"]); w = sum(tot["WRITE_SIZE"]) / len(tot["WRITE_SIZE"])
        print(lib, "HBM bytes per conv launch: %.1f MB (fetch x2 corrected %.1f + write %.1f); algorithmic 169.6 MB" % ((2 * f + w) * 1024 / 1e6, 2 * f * 1024 / 1e6, w * 1024 / 1e6))
PY
