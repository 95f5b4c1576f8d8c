"""Persistent tile loop (gemm6.hip, mc_gemm_tileloop_f16) next to gemm5 (one 256x320 tile per workgroup) and the vendor GEMM
(torch.matmul -> hipBLASLt) on the wide-N short-K Linear shapes of a config-2 CFG-batch forward and their neighbours.

Per shape, arms interleaved in one process, >= --window seconds each, random fp16 operands, TFLOP/s from HIP events over the
window, package power / shader clock from rocm-smi samples inside it (tools/vendor_anchor.py's sampler):

  vendor | gemm5 (cfg 11) | tile loop static | tile loop dynamic | tile loop dynamic, stores drained (flag 0x1)

"as used" = with the layer's own epilogue (bias / fused GEGLU / residual).  Before timing, every tile-loop arm is compared
with gemm5 bit for bit (the outputs must be EQUAL), repeated --reps times, plain and as used; and the counter blocks must be
zero again afterwards.  One JSON line per (shape, arm), a markdown table at the end.

  python tools/tileloop_bench.py [--window 1.0] [--only ff1] [--md gpurun_out/r06_tileloop.md]
"""
import argparse
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from motionclone_amd import lib, ops  # noqa: E402
from tools.vendor_anchor import Smi, rnd, run_window  # noqa: E402

dev = torch.device("cuda:0")

# name, M, N, K, bias, residual, geglu
SHAPES = [
    ("ff1_l1 geglu", 32768, 5120, 640, True, False, True),
    ("ff1_l2 geglu", 8192, 10240, 1280, True, False, True),
    ("qkv_l2", 8192, 3840, 1280, False, False, False),
    ("qkv_l1", 32768, 1920, 640, False, False, False),
    ("attn_out_l2 +R", 8192, 1280, 1280, True, True, False),
    ("ff2_l1 +R", 32768, 640, 2560, True, True, False),
    ("attn_out_l1 +R", 32768, 640, 640, True, True, False),
    ("ff1_l0 geglu (K=320)", 131072, 2560, 320, True, False, True),
    ("ff2_l0 +R", 131072, 320, 1280, True, True, False),
    ("ff1_l2 B=1 geglu", 4096, 10240, 1280, True, False, True),
    ("qkv_l2 B=1", 4096, 3840, 1280, False, False, False),
    # the 8x8 level (and the backward of the 16x16 one): few tiles - what split-K + reduce / the small tiles serve today
    ("small qkv_l3", 2048, 3840, 1280, False, False, False),
    ("small proj_l3 +R", 2048, 1280, 1280, True, True, False),
    ("small ff1_l3 geglu", 2048, 10240, 1280, True, False, True),
    ("small ff2_l3 +R", 2048, 1280, 5120, True, True, False),
    ("small ff2_l2 bwd", 4096, 5120, 1280, False, False, False),
    ("small proj_l2 bwd", 4096, 1280, 1280, False, False, False),
    ("small ff1_l2 bwd (K=5120)", 4096, 1280, 5120, False, False, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--window", type=float, default=1.0)
    ap.add_argument("--only", default="")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--md", default="gpurun_out/r06_tileloop.md")
    a = ap.parse_args()
    lib.load()
    smi = Smi()
    smi.start()
    rows = []
    print(json.dumps(dict(device=torch.cuda.get_device_name(0), torch=torch.__version__, window_s=a.window)), flush=True)
    for name, M, N, K, has_b, has_r, geglu in SHAPES:
        if a.only and a.only not in name:
            continue
        flop = 2.0 * M * N * K
        x, w = rnd(M, K, seed=1), rnd(N, K, s=0.02, seed=2)
        wt = w.t()
        bias = (torch.randn(1, N, device=dev, generator=torch.Generator(device=dev).manual_seed(5)) * 0.1) if has_b else None
        R = rnd(M, N, seed=3) if has_r else None
        nout = N // 2 if geglu else N
        o_v = torch.empty((M, N), dtype=torch.float16, device=dev)
        o_ref = torch.empty((M, nout), dtype=torch.float16, device=dev)
        o_t = torch.empty((M, nout), dtype=torch.float16, device=dev)
        kw = dict(bias=bias, residual=R, geglu=geglu)

        # ---- bit-identity with gemm5, plain and as used, every variant, repeated --------------------------------------
        ident = {}
        for label, kws in (("plain", {}), ("as used", kw)):
            if label == "plain":
                ref = torch.empty((M, N), dtype=torch.float16, device=dev)
                got = torch.empty((M, N), dtype=torch.float16, device=dev)
            else:
                ref, got = o_ref, o_t
            ops.gemm(x, w, cfg=11, out=ref, **kws)
            for dyn, strict in ((False, False), (True, False), (True, True)):
                bad = 0
                for _ in range(a.reps):
                    got.fill_(float("nan"))
                    r = ops.gemm_tileloop(x, w, out=got, dynamic=dyn, strict_order=strict, **kws)
                    if r is None:
                        bad = -1
                        break
                    bad += int((got != ref).sum().item()) + int(torch.isnan(got).sum().item())
                ident["%s %s%s" % (label, "dynamic" if dyn else "static", " strict" if strict else "")] = bad
            # stream-K: cut tiles are summed piecewise - compared with a tolerance (one fp16 ulp of the largest element)
            worst = 0.0
            for _ in range(a.reps):
                got.fill_(float("nan"))
                r = ops.gemm_tileloop(x, w, out=got, stream_k=True, **kws)
                if r is None:
                    worst = -1.0
                    break
                d = (got.float() - ref.float()).abs().max().item()
                worst = float("nan") if d != d else max(worst, d)
            ident["%s stream-K max|diff| (ref max %.3g)" % (label, ref.float().abs().max().item())] = worst
            del ref, got
        torch.cuda.synchronize()
        dirty = sum(int(v.count_nonzero().item()) for v in ops._tile_slabs.values())
        print(json.dumps(dict(shape=name, M=M, N=N, K=K, unequal_elements_vs_gemm5=ident, counter_words_nonzero_after=dirty)),
              flush=True)

        iters = max(4, int(3000.0 / max(20.0, flop / 1.0e9)))
        arms = [
            ("vendor (plain product)", lambda: torch.matmul(x, wt, out=o_v)),
            ("gemm5 as used", lambda: ops.gemm(x, w, cfg=11, out=o_ref, **kw)),
            ("tile loop static", lambda: ops.gemm_tileloop(x, w, out=o_t, dynamic=False, **kw)),
            ("tile loop dynamic", lambda: ops.gemm_tileloop(x, w, out=o_t, dynamic=True, **kw)),
            ("tile loop dynamic, stores drained", lambda: ops.gemm_tileloop(x, w, out=o_t, dynamic=True, strict_order=True, **kw)),
            ("tile loop stream-K", lambda: ops.gemm_tileloop(x, w, out=o_t, stream_k=True, **kw)),
            ("library's own choice (ops.gemm)", lambda: ops.gemm(x, w, out=o_t, **kw)),
            # the bare product (what the vendor arm computes): tools/vendor_anchor.py's comparison
            ("gemm5 plain", lambda: ops.gemm(x, w, cfg=11, out=o_v)),
            ("tile loop dynamic plain", lambda: ops.gemm_tileloop(x, w, out=o_v, dynamic=True)),
            ("tile loop stream-K plain", lambda: ops.gemm_tileloop(x, w, out=o_v, stream_k=True)),
        ]
        for arm, fn in arms:
            if fn() is None:
                print(json.dumps(dict(shape=name, arm=arm, refused=True)), flush=True)
                continue
            us, t0, t1, n = run_window(fn, a.window, iters)
            watts, mhz, ns = smi.window(t0, t1)
            row = dict(shape=name, M=M, N=N, K=K, arm=arm, us=round(us, 2), TFLOPs=round(flop / us / 1e6, 1), calls=n,
                       watts_median=watts, sclk_mhz_median=mhz,
                       mc_kernel=lib.load().mc_gemm_last_kernel() if not arm.startswith("vendor") else None)
            rows.append(row)
            print(json.dumps(row), flush=True)
            time.sleep(0.2)
        del x, w, R, o_v, o_ref, o_t
        torch.cuda.empty_cache()
    smi.stop = True
    with open(a.md, "w") as f:
        f.write("| shape (M x N x K) | arm | us | TFLOP/s | vs vendor | vs gemm5 | W | MHz | kernel |\n|---|---|---|---|---|---|---|---|---|\n")
        base = {}
        for r in rows:
            if r["arm"].startswith("vendor"):
                base[(r["shape"], "v")] = r["us"]
            if r["arm"].startswith("gemm5"):
                base[(r["shape"], "g")] = r["us"]
        for r in rows:
            v, g = base.get((r["shape"], "v")), base.get((r["shape"], "g"))
            f.write("| %s (%d x %d x %d) | %s | %.1f | %.0f | %s | %s | %s | %s | %s |\n" % (
                r["shape"], r["M"], r["N"], r["K"], r["arm"], r["us"], r["TFLOPs"],
                "%.3f" % (v / r["us"]) if v else "-", "%.3f" % (g / r["us"]) if g else "-",
                r["watts_median"], r["sclk_mhz_median"], r["mc_kernel"] if r["mc_kernel"] is not None else "hipBLASLt"))


if __name__ == "__main__":
    main()
