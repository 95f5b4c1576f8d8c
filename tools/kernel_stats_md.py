"""rocprofv3 --kernel-trace --stats CSVs of tools/gpu_profile_*_final.sh -> profiles/<round>_kernel_stats.md (+ the two CSVs).
usage: python tools/kernel_stats_md.py <prof_a dir> <prof_b dir> <out md> <round tag> [videos_a videos_b]"""
import csv
import glob
import json
import os
import re
import shutil
import sys


def stats_csv(d):
    fs = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    assert fs, "no kernel_stats.csv under " + d
    return max(fs, key=os.path.getsize)


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("mc::", "")
    name = re.sub(r"^_ZN2mc", "", name)
    return name[:72]


def bench_value(d):
    try:
        lines = [l for l in open(os.path.join(d, "bench.json")) if l.startswith("{")]
        return json.loads(lines[-1])["value"]
    except Exception:
        return None


def bench_packing(d):
    """(lanes, videos batched per lane) of the traced bench run, from its line's config"""
    try:
        lines = [l for l in open(os.path.join(d, "bench.json")) if l.startswith("{")]
        c = json.loads(lines[-1])["config"]
        return int(c.get("videos_in_flight_per_gpu", 3)), int(c.get("videos_batched_per_lane", 1))
    except Exception:
        return 3, 1


def table(path, videos, top=34):
    rows = list(csv.DictReader(open(path)))
    ncalls = sum(int(r["Calls"]) for r in rows)
    tot = sum(int(r["TotalDurationNs"]) for r in rows) / 1e9
    mine = [r for r in rows if "mc::" in r["Name"] or "_ZN2mc" in r["Name"]]
    out = ["| kernel | calls | calls / video | total ms | avg us | % |", "|---|---|---|---|---|---|"]
    for r in rows[:top]:
        out.append("| `%s` | %d | %d | %.1f | %.1f | %s |" % (short(r["Name"]), int(r["Calls"]), round(int(r["Calls"]) / videos),
                                                          int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
    return ncalls, tot, sum(int(r["Calls"]) for r in mine), "\n".join(out)


def durations_json(stats_path, trace_dir, videos, vpm, tag):
    """-> profiles/kernel_durations_timed.json (read by bench.py: roofline_timed): per kernel calls / average duration of the
    TIMED regime, the summed kernel time, and - when the raw kernel trace is still there - the overlap factor = summed kernel
    time / length of the union of the kernel intervals (how many kernels run at once on average)."""
    rows = list(csv.DictReader(open(stats_path)))
    kern = {re.sub(r"^void ", "", re.sub(r"\(.*$", "", r["Name"])).replace("mc::", ""):
            dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3) for r in rows}
    total = sum(int(r["TotalDurationNs"]) for r in rows) / 1e9
    overlap = None
    tr = glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True)
    if tr:
        iv = []
        for r in csv.DictReader(open(max(tr, key=os.path.getsize))):
            iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        iv.sort()
        busy, cur_s, cur_e = 0, None, None
        for s_, e_ in iv:
            if cur_e is None or s_ > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = s_, e_
            else:
                cur_e = max(cur_e, e_)
        if cur_e is not None:
            busy += cur_e - cur_s
        overlap = sum(e_ - s_ for s_, e_ in iv) / max(1, busy)
    try:
        import subprocess
        code = subprocess.run(["git", "rev-parse", "--short", "HEAD"], stdout=subprocess.PIPE, text=True).stdout.strip() or None
    except Exception:   # noqa: BLE001
        code = None
    # what bench.py's roofline_timed checks: the source stamp of the library the traced run loaded (build.py writes it next to the
    # .so) and the GPU it ran on - a trace of other code or another device is refused there
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        lib_stamp = open(os.path.join(here, "motionclone_amd", "csrc", "libmotionclone_hip.so.stamp")).read().strip()
    except OSError:
        lib_stamp = None
    try:
        import torch
        device = torch.cuda.get_device_name(0) if torch.cuda.is_available() else None
    except Exception:   # noqa: BLE001
        device = None
    lanes, batch = bench_packing(trace_dir)
    return dict(lib_stamp=lib_stamp, device=device, lanes=lanes, batch=batch, regime="%d lanes x %d videos in flight, no probe videos: `%s` under rocprofv3 --kernel-trace --stats (%s)"
                       % (lanes, batch, os.environ.get("MC_PROFILE_CMD_A", "python bench.py --no-cpu-baseline --no-vae --no-probe --steps 6 --warmup 3"), tag), videos_in_trace=videos, videos_per_min_under_profiler=vpm,
                total_kernel_s=total, overlap=overlap, code=code, kernels=kern)


a_dir, b_dir, out_md, tag = sys.argv[1:5]
va = int(sys.argv[5]) if len(sys.argv) > 5 else 10   # 3 warm-up + 3 timed + 4 eager (probe pass)
vb = int(sys.argv[6]) if len(sys.argv) > 6 else 7    # 1 warm-up + 2 timed + 4 eager
pa, pb = stats_csv(a_dir), stats_csv(b_dir)
base = out_md[:-3]
shutil.copy(pa, base + "_a.csv")
shutil.copy(pb, base + "_b.csv")
na, ta, ma, tab_a = table(pa, va)
nb, tb, mb, tab_b = table(pb, vb)
cmd_a = os.environ.get("MC_PROFILE_CMD_A", "python bench.py --no-cpu-baseline --no-vae --no-detail --no-probe --steps 6 --warmup 3")
cmd_b = os.environ.get("MC_PROFILE_CMD_B", "python bench.py --no-cpu-baseline --no-vae --no-detail --no-probe --no-graphs --inflight 1 --batch 5 --gemm-lanes 2 --steps 5 --warmup 5")
lanes_a, batch_a = bench_packing(a_dir)
md = """# rocprofv3 --kernel-trace --stats of the final %s code (1x MI355X)

`tools/gpu_profile.sh` + `tools/kernel_stats_md.py`.  Launch counts per video divide by ALL videos in the trace: (a) warm-up +
timed videos only (`--no-probe`: the trace holds the timed regime's kernels and nothing else; it is what
`profiles/kernel_durations_timed.json` / the bench line's `roofline_timed` are made of), (b) warm-up, timed and the eager videos of
the bench's probe pass.  Kernel names: `gemm5_kernel<MODE, EPI, VAR, BM, BN, waves, ring stages, RES, GNS>` (MODE 0 dense, 1
conv3x3, 2 stride-2, 3 upsample+conv, 4 transposed; EPI 1 = fused GEGLU; RES 1 = residual in the epilogue; GNS 1 = the epilogue
leaves the GroupNorm statistics of its output), `gemm6_kernel<EPI, RES, VAR, SK>` = the persistent tile loop over 256x320 tiles,
`gemm4_kernel<20, GEGLU, NORM>` = K = 320 streaming kernel (NORM 1 = LayerNorm, 2 = GroupNorm applied to the rows in registers),
`attn_*_ring_kernel<DT, rows/16 per wave>` = the LDS-DMA ring attention (DT 3: d = 40, 5: d = 80).

## (a) `%s`: %d lanes x %d videos batched per lane (kernel durations are measured while kernels of the other lane share the CUs)

Bench line of this run: **%s videos/min** (under the profiler); videos in the trace (warm-up + timed): **%d**; %d kernel launches
= **%d per video** (%d of them this library's); total kernel time %.1f s.

%s

## (b) `%s`: ONE lane's job alone on the eager launch sequence (the regime of the roofline probe of bench.py: `roofline.avg_launch_us` is measured there with HIP events)

Bench line of this run: **%s videos/min**; videos in the trace (warm-up + timed) = **%d**; %d kernel launches = **%d per video**;
total kernel time %.1f s = %.2f s per video.

%s
""" % (tag, cmd_a, lanes_a, batch_a, bench_value(a_dir), va, na, round(na / va), round(ma / va), ta, tab_a, cmd_b, bench_value(b_dir), vb, nb, round(nb / vb), tb, tb / vb, tab_b)
open(out_md, "w").write(md)
json.dump(durations_json(pa, a_dir, va, bench_value(a_dir), tag), open(os.path.join(os.path.dirname(out_md), "kernel_durations_timed.json"), "w"), indent=1)
print(out_md, "launches per video:", round(na / va), round(nb / vb))
