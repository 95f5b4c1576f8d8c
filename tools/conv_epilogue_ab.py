"""A/B of gemm5's epilogue on the 3x3 convolutions and 128-row dense tiles AS THE ENGINE CALLS THEM (bias rows per sample from the
time embedding, residual): this process loads whatever MC_HIP_LIB points to, so run it twice -

  python tools/conv_epilogue_ab.py                                              # the product library (round-6 epilogue)
  MC_HIP_LIB=tools/_build/libmotionclone_hip_oldepi.so python tools/conv_epilogue_ab.py   # round 5's epilogue

One JSON line per shape: median of --rounds x --iters launches, HIP events."""
import argparse
import json
import statistics
import sys

import torch

sys.path.insert(0, ".")
from motionclone_amd import lib, ops  # noqa: E402

dev = torch.device("cuda:0")
F2 = 32


def r(*shape, s=1.0, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(*shape, device=dev, generator=g) * s).half()


# name, mode, M, N, K, residual, per-sample bias, geom
SHAPES = [
    ("conv_l0 320->320 +temb", 1, F2 * 4096, 320, 2880, False, True, (64, 64, 64, 64)),
    ("conv_l0 320->320 +R", 1, F2 * 4096, 320, 2880, True, False, (64, 64, 64, 64)),
    ("conv_l0 640->320 +temb", 1, F2 * 4096, 320, 5760, False, True, (64, 64, 64, 64)),
    ("conv_l1 640->640 +temb", 1, F2 * 1024, 640, 5760, False, True, (32, 32, 32, 32)),
    ("conv_l1 640->640 +R", 1, F2 * 1024, 640, 5760, True, False, (32, 32, 32, 32)),
    ("conv_l1 1280->640 +temb", 1, F2 * 1024, 640, 11520, False, True, (32, 32, 32, 32)),
    ("conv_l2 1280->1280 +R", 1, F2 * 256, 1280, 11520, True, False, (16, 16, 16, 16)),
    ("conv_up l1->l0 640", 3, F2 * 4096, 640, 5760, False, False, (32, 32, 64, 64)),
    ("dense 128-row attn_out_l2 +R", 0, 8192, 1280, 1280, True, False, None),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=8)
    a = ap.parse_args()
    lib.load()
    ops.TILELOOP = False
    print(json.dumps(dict(library=lib.HIP_LIB_PATH)), flush=True)
    for name, mode, M, N, K, res, temb, geom in SHAPES:
        if mode == 0:
            x = r(M, K, seed=1)
            kw = dict(cfg=15)
        else:
            Hs, Ws, Ho, Wo = geom
            frames = M // (Ho * Wo)
            x = r(frames * Hs * Ws, K // 9, seed=1)
            kw = dict(mode=mode, geom=geom, m_out=M)
        w = r(N, K, s=0.02, seed=2)
        R = r(M, N, seed=3) if res else None
        if temb:
            bias, rpb = torch.randn(2, N, device=dev), M // 2
        else:
            bias, rpb = torch.randn(1, N, device=dev), 0
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        fn = lambda: ops.gemm(x, w, bias=bias, rows_per_batch=rpb, residual=R, out=out, **kw)   # noqa: E731
        fn()
        torch.cuda.synchronize()
        ts = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(a.rounds):
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(1e3 * e0.elapsed_time(e1) / a.iters)
        us = statistics.median(ts)
        print(json.dumps(dict(shape=name, M=M, N=N, K=K, us=round(us, 1), TFLOPs=round(2.0 * M * N * K / us / 1e6, 0),
                              kernel=lib.load().mc_gemm_last_kernel(), checksum=float(out.float().abs().mean()))), flush=True)


if __name__ == "__main__":
    main()
