#!/bin/bash
# Package power and shader clock while a bench command runs: rocm-smi sampled every 0.5 s into gpurun_out/<tag>_smi.csv
# (columns as `rocm-smi --showclocks --showpower --csv` prints them), summary of the samples taken while the GPU was busy.
#   bash tools/smi_power.sh <tag> <bench args...>      (run with MC_HIP_LIB / MC_* set for A/B builds)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
tag=$1; shift
mkdir -p gpurun_out
( while true; do rocm-smi --showclocks --showpower --csv 2>/dev/null | grep '^card0' ; sleep 0.5; done ) > gpurun_out/${tag}_smi.csv &
SMI=$!
python bench.py "$@" > gpurun_out/${tag}_bench.log 2>&1
kill $SMI 2>/dev/null
python - "$tag" <<'PY'
import re, sys, json
tag = sys.argv[1]
rows = []
for l in open("gpurun_out/%s_smi.csv" % tag):
    f = l.strip().split(",")
    try:
        mhz = [int(m) for m in re.findall(r"\((\d+)Mhz\)", l)]
        rows.append((max(mhz[2:4]) if len(mhz) >= 4 else mhz[-1], float(f[-1])))
    except Exception:
        pass
busy = [r for r in rows if r[1] > 600]
line = [x for x in open("gpurun_out/%s_bench.log" % tag) if x.startswith("{")]
v = json.loads(line[-1])["value"] if line else None
if busy:
    busy.sort(key=lambda r: r[1])
    w = [r[1] for r in busy]; c = sorted(r[0] for r in busy)
    print(json.dumps(dict(tag=tag, videos_per_min=v, samples_busy=len(busy), watts_median=w[len(w) // 2], watts_p90=w[int(0.9 * len(w))],
                          sclk_mhz_median=c[len(c) // 2], sclk_mhz_p10=c[int(0.1 * len(c))])))
else:
    print(json.dumps(dict(tag=tag, videos_per_min=v, samples=len(rows))))
PY
