"""256x320 tiles with EIGHT waves (64 x 160 wave tiles, two waves per SIMD: cfg 11) vs FOUR waves (128 x 160 wave tiles, one wave per
SIMD with 256 + 256 registers: cfg 8) vs the persistent tile loop, as the engine calls the layers, interleaved rounds in one
process; outputs must be EQUAL.  One JSON line per shape.   python tools/wave_tile_ab.py [--rounds 5] [--iters 8]"""
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


# name, mode, M, N, K, bias rows (0 none, 1 one, 2 per sample), residual, geglu, geom
SHAPES = [
    ("ff1_l1 geglu", 0, 32768, 5120, 640, 1, False, True, None),
    ("ff1_l1 plain", 0, 32768, 5120, 640, 0, False, False, None),
    ("ff1_l2 plain", 0, 8192, 10240, 1280, 0, False, False, None),
    ("attn_out_l2 +R", 0, 8192, 1280, 1280, 1, True, False, None),
    ("ff2_l2 +R", 0, 8192, 1280, 5120, 1, True, False, None),
    ("ff1_l2 geglu", 0, 8192, 10240, 1280, 1, False, True, None),
    ("qkv_l2", 0, 8192, 3840, 1280, 0, False, False, None),
    ("qkv_l1", 0, 32768, 1920, 640, 0, False, False, None),
    ("ff2_l1 +R", 0, 32768, 640, 2560, 1, True, False, None),
    ("attn_out_l1 +R", 0, 32768, 640, 640, 1, True, False, None),
    ("ff2_l0 +R", 0, 131072, 320, 1280, 1, True, False, None),
    ("conv_l0 320->320 +temb", 1, F2 * 4096, 320, 2880, 2, False, False, (64, 64, 64, 64)),
    ("conv_l0 320->320 +R", 1, F2 * 4096, 320, 2880, 1, True, False, (64, 64, 64, 64)),
    ("conv_l0 640->320 +temb", 1, F2 * 4096, 320, 5760, 2, False, False, (64, 64, 64, 64)),
    ("conv_l1 640->640 +R", 1, F2 * 1024, 640, 5760, 1, True, False, (32, 32, 32, 32)),
    ("conv_l1 1280->640 +temb", 1, F2 * 1024, 640, 11520, 2, False, False, (32, 32, 32, 32)),
    ("conv_l2 1280->1280 +R", 1, F2 * 256, 1280, 11520, 1, True, False, (16, 16, 16, 16)),
    ("conv_up l1->l0 640", 3, F2 * 4096, 640, 5760, 1, False, False, (32, 32, 64, 64)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--dense-only", action="store_true")
    ap.add_argument("--scale-m", type=int, default=1, help="V videos batched per lane: V times the rows of every shape")
    a = ap.parse_args()
    lib.load()
    for name, mode, M, N, K, nb, res, geglu, geom in SHAPES:
        if a.dense_only and mode != 0:
            continue
        M *= a.scale_m
        if mode == 0:
            x = r(M, K, seed=1)
            kw = {}
        else:
            Hs, Ws, Ho, Wo = geom
            x = r((M // (Ho * Wo)) * Hs * Ws, K // 9, seed=1)
            kw = dict(mode=mode, geom=geom, m_out=M)
        w = r(N, K, s=0.02, seed=2)
        R = r(M, N, seed=3) if res else None
        bias = torch.randn(nb, N, device=dev) if nb else None
        rpb = M // 2 if nb == 2 else 0
        nout = N // 2 if geglu else N
        arms = {"8 waves (cfg 11)": dict(cfg=11), "4 waves (cfg 8)": dict(cfg=8)}
        if mode == 0:
            arms["tile loop"] = dict(tileloop=True)
            arms["tile loop, column blocks outermost"] = "col"
            arms["256x256, 4 waves, 128x128 wave tiles (cfg 7)"] = dict(cfg=7)
            if not geglu and not res:
                arms["vendor (torch.matmul)"] = None
        outs = {k: torch.empty((M, nout), dtype=torch.float16, device=dev) for k in arms}
        wt = w.t().contiguous()
        fns = {k: ((lambda k=k: ops.gemm_tileloop(x, w, bias=bias, rows_per_batch=rpb, residual=R, geglu=geglu, out=outs[k], col_outer=True))
                   if v == "col" else
                   (lambda k=k, v=v: ops.gemm(x, w, bias=bias, rows_per_batch=rpb, residual=R, geglu=geglu, out=outs[k], **kw, **v))
                   if v is not None else (lambda k=k: torch.matmul(x, wt, out=outs[k])))
               for k, v in arms.items()}
        for f in fns.values():
            f()
        torch.cuda.synchronize()
        ref = outs["8 waves (cfg 11)"]
        equal = {k: bool(torch.equal(o, ref)) for k, o in outs.items()}
        ts = {k: [] for k in arms}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(a.rounds):
            for k, f in fns.items():
                e0.record()
                for _ in range(a.iters):
                    f()
                e1.record()
                torch.cuda.synchronize()
                ts[k].append(1e3 * e0.elapsed_time(e1) / a.iters)
        row = dict(shape=name, M=M, N=N, K=K)
        for k in arms:
            us = statistics.median(ts[k])
            row[k] = dict(us=round(us, 1), TFLOPs=round(2.0 * M * N * K / us / 1e6), equal_to_8_waves=equal[k])
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
