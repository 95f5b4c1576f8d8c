"""Vendor-library anchor for the "power wall" (round-4 verdict, item 2): what does the MI355X sustain on THIS library's GEMM
shapes when the vendor's own fp16 GEMM (torch.matmul -> hipBLASLt / rocBLAS) runs them, next to this library's kernels, on the
same box, in the same process, with package power and shader clock sampled (rocm-smi) during every window?

For each shape (the eight dense shapes of profiles/r04_pmc_hbm_traffic_three_lanes.md + the two deep-K 3x3 convolutions as
the dense GEMMs of their im2col form), one window of >= `--window` seconds per arm:

  vendor  random | vendor  zeros | mc plain random | mc plain zeros | mc as used (residual / GEGLU epilogue) random

"zeros" = all-zero operands: the same instruction stream with (almost) no switching activity - MI355X_MICROARCH.md (DVFS
give-back) predicts the clock, and with it the TFLOP/s, to rise when the kernel is power- and not issue-limited.  Arms are
interleaved per shape (cdna guide 5.4 rule 24); TFLOP/s from HIP events over the window, watts / MHz = median of the rocm-smi
samples whose timestamp falls inside the window.  One JSON line per (shape, arm) + a markdown table at the end.

  python tools/vendor_anchor.py [--window 2.5] [--only ff1] > gpurun_out/r05_vendor_anchor.jsonl
"""
import argparse
import json
import re
import statistics
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
from motionclone_amd import lib, ops  # noqa: E402

dev = torch.device("cuda:0")

# name, M, N, K, residual, geglu, conv geometry of the as-used arm (None = dense)
SHAPES = [
    ("qkv_l1", 32768, 1920, 640, False, False, None),
    ("attn_out_l1 +R", 32768, 640, 640, True, False, None),
    ("ff1_l1 geglu", 32768, 5120, 640, False, True, None),
    ("ff2_l1 +R", 32768, 640, 2560, True, False, None),
    ("qkv_l2", 8192, 3840, 1280, False, False, None),
    ("attn_out_l2 +R", 8192, 1280, 1280, True, False, None),
    ("ff1_l2 geglu", 8192, 10240, 1280, False, True, None),
    ("ff2_l0 +R", 131072, 320, 1280, True, False, None),
    ("conv_l1 1280->640 (im2col K)", 32768, 640, 11520, False, False, (32, 32, 32, 32)),
    ("conv_l0 640->320 (im2col K)", 131072, 320, 5760, False, False, (64, 64, 64, 64)),
]


class Smi(threading.Thread):
    """rocm-smi sampled in a loop (the invocation itself takes ~0.2-0.4 s): (time, sclk MHz, watts) rows"""

    def __init__(self):
        super().__init__(daemon=True)
        self.rows = []
        self.stop = False

    def run(self):
        while not self.stop:
            t = time.time()
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], stdout=subprocess.PIPE,
                                     stderr=subprocess.DEVNULL, text=True, timeout=5).stdout
            except Exception:   # noqa: BLE001
                time.sleep(0.5)
                continue
            for line in out.splitlines():
                if line.startswith("card0"):
                    mhz = [int(m) for m in re.findall(r"\((\d+)Mhz\)", line)]
                    try:
                        w = float(line.strip().split(",")[-1])
                    except ValueError:
                        continue
                    sclk = max(mhz[2:4]) if len(mhz) >= 4 else (mhz[-1] if mhz else 0)
                    self.rows.append((0.5 * (t + time.time()), sclk, w))
            time.sleep(0.05)

    def window(self, t0, t1):
        sel = [r for r in self.rows if t0 + 0.3 <= r[0] <= t1]     # the first 0.3 s: ramp
        if not sel:
            return None, None, 0
        return statistics.median(r[2] for r in sel), statistics.median(r[1] for r in sel), len(sel)


def rnd(*shape, s=1.0, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(*shape, device=dev, generator=g) * s).half()


def run_window(fn, seconds, iters):
    """call fn() back to back for >= seconds; -> (us per call from HIP events over the whole window, wall t0, wall t1)"""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    t0 = time.time()
    e0.record()
    while True:
        for _ in range(iters):
            fn()
        n += iters
        torch.cuda.synchronize()     # one sync per batch of launches: the queue never runs dry for long (iters x ~100 us)
        if time.time() - t0 >= seconds:
            break
    e1.record()
    torch.cuda.synchronize()
    t1 = time.time()
    return 1e3 * e0.elapsed_time(e1) / n, t0, t1, n


def shared(fn, lanes=3):
    ops.set_gemm_share(lanes)
    try:
        return fn()
    finally:
        ops.set_gemm_share(1)


# Pricing of a fused FeedForward kernel (LN -> Linear -> GEGLU -> Linear + residual, hidden never in HBM) with EXISTING kernels:
# such a kernel keeps its output accumulators for 128 rows and streams W1 / W2 past them, i.e. its main loops have the operand
# intensity of the 128-row tile (A tile of 128 rows + a 320-row weight tile per k-step) - gemm5 with BM = 128 (cfg 15) is that
# loop, gemm5 with BM = 256 (cfg 11) is what the two separate launches run today.
FF_SHAPES = [
    ("ff1_l1 geglu", 32768, 5120, 640, False, True),
    ("ff2_l1 +R", 32768, 640, 2560, True, False),
    ("ff1_l2 geglu", 8192, 10240, 1280, False, True),
    ("ff2_l2 +R", 8192, 1280, 5120, True, False),
    ("ff2_l0 +R", 131072, 320, 1280, True, False),
]


def ff_pricing(a, smi):
    out = []
    for name, M, N, K, res, geglu in FF_SHAPES:
        flop = 2.0 * M * N * K
        x, w = rnd(M, K, seed=1), rnd(N, K, s=0.02, seed=2)
        R = rnd(M, N, seed=3) if res else None
        o = torch.empty((M, N // 2 if geglu else N), dtype=torch.float16, device=dev)
        iters = max(4, int(3000.0 / max(20.0, flop / 1.0e9)))
        for cfg, label in ((11, "256-row tiles (today)"), (15, "128-row tiles (the fused kernel's operand intensity)")):
            fn = lambda: ops.gemm(x, w, residual=R, geglu=geglu, cfg=cfg, out=o)   # noqa: E731
            try:
                us, t0, t1, n = run_window(fn, a.window, iters)
            except RuntimeError as e:
                print(json.dumps(dict(ff_pricing=name, cfg=cfg, error=str(e))), flush=True)
                continue
            watts, mhz, ns = smi.window(t0, t1)
            row = dict(ff_pricing=name, M=M, N=N, K=K, cfg=cfg, tiles=label, us=round(us, 2), TFLOPs=round(flop / us / 1e6, 1),
                       hidden_round_trip_MB=round(2 * M * (N // 2 if geglu else K) * 2 / 1e6, 1) if (geglu or name.startswith("ff2")) else None,
                       watts_median=watts, sclk_mhz_median=mhz)
            out.append(row)
            print(json.dumps(row), flush=True)
        del x, w, R, o
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ff-pricing", action="store_true", help="also time the FeedForward shapes on 256- vs 128-row tiles")
    ap.add_argument("--window", type=float, default=2.5)
    ap.add_argument("--only", default="")
    ap.add_argument("--md", default="gpurun_out/r05_vendor_anchor.md")
    a = ap.parse_args()
    lib.load()
    smi = Smi()
    smi.start()
    try:
        blas = str(torch.backends.cuda.preferred_blas_library())
    except Exception:   # noqa: BLE001
        blas = "unknown"
    rows = []
    print(json.dumps(dict(device=torch.cuda.get_device_name(0), torch=torch.__version__, preferred_blas_library=blas,
                          window_s=a.window)), flush=True)
    time.sleep(1.0)
    for name, M, N, K, res, geglu, geom in SHAPES:
        if a.only and a.only not in name:
            continue
        flop = 2.0 * M * N * K
        x_r, w_r = rnd(M, K, seed=1), rnd(N, K, s=0.02, seed=2)
        x_z, w_z = torch.zeros_like(x_r), torch.zeros_like(w_r)
        R = rnd(M, N, seed=3) if res else None
        o_v = torch.empty((M, N), dtype=torch.float16, device=dev)
        o_m = torch.empty((M, N), dtype=torch.float16, device=dev)
        o_g = torch.empty((M, N // 2), dtype=torch.float16, device=dev) if geglu else None
        wt_r, wt_z = w_r.t(), w_z.t()
        iters = max(4, int(3000.0 / max(20.0, flop / 1.0e9)))      # ~3 ms of launches per sync at 1 PFLOP/s

        arms = [
            ("vendor random", lambda: torch.matmul(x_r, wt_r, out=o_v)),
            ("vendor zeros", lambda: torch.matmul(x_z, wt_z, out=o_v)),
            ("mc plain random", lambda: ops.gemm(x_r, w_r, out=o_m)),
            ("mc plain zeros", lambda: ops.gemm(x_z, w_z, out=o_m)),
            # the tile choice of the TIMED regime (three launch sequences in flight: 256 x 320 tiles everywhere)
            ("mc plain random, three-lane tile choice", lambda: shared(lambda: ops.gemm(x_r, w_r, out=o_m))),
        ]
        if res:
            arms.append(("mc as used (+R) random", lambda: ops.gemm(x_r, w_r, residual=R, out=o_m)))
        if geglu:
            arms.append(("mc as used (GEGLU) random", lambda: ops.gemm(x_r, w_r, geglu=True, out=o_g)))
        if geom is not None:
            # the conv itself: activations [frames * Hs * Ws, Cin], packed 3x3 weights - the arm the im2col GEMM stands in for
            Hs, Ws, Ho, Wo = geom
            cin = K // 9
            xc = rnd(M, cin, seed=4)
            arms.append(("mc conv3x3 random", lambda: ops.gemm(xc, w_r, mode=ops.CONV_S1, geom=geom, m_out=M, out=o_m)))
        # correctness of the comparison: both libraries compute the same product
        torch.matmul(x_r, wt_r, out=o_v)
        ops.gemm(x_r, w_r, out=o_m)
        torch.cuda.synchronize()
        rel = ((o_v.float() - o_m.float()).norm() / o_v.float().norm().clamp_min(1e-9)).item()
        for arm, fn in arms:
            us, t0, t1, n = run_window(fn, a.window, iters)
            watts, mhz, ns = smi.window(t0, t1)
            row = dict(shape=name, M=M, N=N, K=K, arm=arm, us=round(us, 2), TFLOPs=round(flop / us / 1e6, 1), calls=n,
                       watts_median=watts, sclk_mhz_median=mhz, smi_samples=ns, rel_l2_vendor_vs_mc=round(rel, 6),
                       mc_kernel=lib.load().mc_gemm_last_kernel() if arm.startswith("mc") else None)
            rows.append(row)
            print(json.dumps(row), flush=True)
            time.sleep(0.3)
        del x_r, w_r, x_z, w_z, R, o_v, o_m, o_g
        torch.cuda.empty_cache()
    ff = ff_pricing(a, smi) if a.ff_pricing else []
    smi.stop = True
    # markdown table
    with open(a.md, "w") as f:
        f.write("| shape (M x N x K) | arm | us | TFLOP/s | frac of 2.5 PF | W (median) | sclk MHz | kernel |\n|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            f.write("| %s (%d x %d x %d) | %s | %.1f | %.0f | %.3f | %s | %s | %s |\n" % (
                r["shape"], r["M"], r["N"], r["K"], r["arm"], r["us"], r["TFLOPs"], r["TFLOPs"] / 2500.0,
                r["watts_median"], r["sclk_mhz_median"], r["mc_kernel"] if r["mc_kernel"] is not None else "hipBLASLt / rocBLAS"))
        if ff:
            f.write("\n| FeedForward shape | tiles | us | TFLOP/s | W | MHz |\n|---|---|---|---|---|---|\n")
            for r in ff:
                f.write("| %s (%d x %d x %d) | %s | %.1f | %.0f | %s | %s |\n" % (r["ff_pricing"], r["M"], r["N"], r["K"], r["tiles"], r["us"],
                                                                             r["TFLOPs"], r["watts_median"], r["sclk_mhz_median"]))


if __name__ == "__main__":
    main()
