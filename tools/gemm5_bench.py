"""A/B of the tiled GEMM / conv kernels on the config-2 (B = 2) shapes: gemm3 (2-stage loop, fp32 staging epilogue) vs gemm5
(4-stage ring, staggered LDS-DMA, wave-private epilogue) and gemm5's schedule experiments, interleaved rounds in ONE process
(cdna guide 5.4 rule 24), random fp16 operands (rule 25).  One JSON line per shape.

  python tools/gemm5_bench.py [--rounds 5] [--iters 10] [--variants 1,11,12,13,14] [--only conv]

variant numbers = mc_gemm_f16 flags bits 12-15: 0 = the library's own choice (incl. split-K), 1 = gemm3 256x320, 4 = gemm3 128x320
(4 waves), 10 = gemm4 (K = 320 only), 11 = gemm5, 12/13/14 = gemm5 with stagger / with the LDS-DMA burst at the top / both,
15 = gemm5 with 128-row tiles."""
import argparse
import json
import statistics
import sys

import torch

sys.path.insert(0, ".")
from motionclone_amd import lib, ops  # noqa: E402

dev = torch.device("cuda:0")


def r(*shape, s=1.0, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(*shape, device=dev, generator=g) * s).half()


# name, mode, M, N, K, residual, geglu, conv geometry (Hs, Ws, Ho, Wo) / None
F2 = 32   # frames x batch of a B = 2 forward
SHAPES = [
    ("qkv_l1", 0, 32768, 1920, 640, False, False, None),
    ("attn_out_l1 +R", 0, 32768, 640, 640, True, False, None),
    ("ff1_l1 geglu", 0, 32768, 5120, 640, False, True, None),
    ("ff2_l1 +R", 0, 32768, 640, 2560, True, False, None),
    ("qkv_l2", 0, 8192, 3840, 1280, False, False, None),
    ("attn_out_l2 +R", 0, 8192, 1280, 1280, True, False, None),
    ("ff1_l2 geglu", 0, 8192, 10240, 1280, False, True, None),
    ("ff2_l2 +R", 0, 8192, 1280, 5120, True, False, None),
    ("ff2_l0 +R", 0, 131072, 320, 1280, True, False, None),
    ("qkv_l0 (K=320)", 0, 131072, 960, 320, False, False, None),
    ("attn_out_l0 +R (K=320)", 0, 131072, 320, 320, True, False, None),
    ("ff1_l0 geglu (K=320)", 0, 131072, 2560, 320, False, True, None),
    ("conv_l0 320->320 +R", 1, F2 * 4096, 320, 2880, True, False, (64, 64, 64, 64)),
    ("conv_l0 640->320", 1, F2 * 4096, 320, 5760, False, False, (64, 64, 64, 64)),
    ("conv_l1 640->640 +R", 1, F2 * 1024, 640, 5760, True, False, (32, 32, 32, 32)),
    ("conv_l1 1280->640", 1, F2 * 1024, 640, 11520, False, False, (32, 32, 32, 32)),
    ("conv_l2 1280->1280 +R", 1, F2 * 256, 1280, 11520, True, False, (16, 16, 16, 16)),
    ("conv_up l1->l0 640", 3, F2 * 4096, 640, 5760, False, False, (32, 32, 64, 64)),
    ("conv_down l0->l1 320", 2, F2 * 1024, 320, 2880, False, False, (64, 64, 32, 32)),
    # the 16x16 (M = 8192 at B = 2, 4096 in the backward) and 8x8 (2048 / 1024) levels: few tiles, deep K
    ("small conv_l2 2560->1280 +R", 1, F2 * 256, 1280, 23040, True, False, (16, 16, 16, 16)),
    ("small conv_l2 bwd 1280->1280", 1, 16 * 256, 1280, 11520, False, False, (16, 16, 16, 16)),
    ("small conv_l3 1280->1280 +R", 1, F2 * 64, 1280, 11520, True, False, (8, 8, 8, 8)),
    ("small conv_l3 2560->1280 +R", 1, F2 * 64, 1280, 23040, True, False, (8, 8, 8, 8)),
    ("small conv_l3 bwd 1280->1280", 1, 16 * 64, 1280, 11520, False, False, (8, 8, 8, 8)),
    ("small qkv_l3", 0, 2048, 3840, 1280, False, False, None),
    ("small proj_l3 +R", 0, 2048, 1280, 1280, True, False, None),
    ("small ff1_l3 geglu", 0, 2048, 10240, 1280, False, True, None),
    ("small ff2_l3 +R", 0, 2048, 1280, 5120, True, False, None),
    ("small proj_l2 bwd", 0, 4096, 1280, 1280, False, False, None),
    ("small qkv_l2 bwd", 0, 4096, 1280, 3840, False, False, None),
    ("small ff2_l2 bwd", 0, 4096, 5120, 1280, False, False, None),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--variants", default="1,11,12,13,14")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    lib.load()
    variants = [int(v) for v in a.variants.split(",")]
    for name, mode, M, N, K, res, geglu, geom in SHAPES:
        if a.only and a.only not in name:
            continue
        if mode == 0:
            x = r(M, K, seed=1)
            kw = {}
        else:
            Hs, Ws, Ho, Wo = geom
            frames = M // (Ho * Wo)
            x = r(frames * Hs * Ws, K // 9, seed=1)
            kw = dict(mode=mode, geom=geom, m_out=M)
        w = r(N, K, s=0.02, seed=2)
        R = r(M, N, seed=3) if res else None
        outs, times = {}, {v: [] for v in variants}
        ok_variants = []

        def call(v, out):   # -1 = the library's own choice with the split-K path kept on gemm3 + splitk_reduce (round 2)
            return ops.gemm(x, w, residual=R, geglu=geglu, cfg=max(v, 0), out=out, g3_splitk=v < 0, **kw)
        for v in variants:
            if geglu and v in (12, 13, 14):
                continue
            if geglu and v in (1, 4, 0) and False:
                continue
            if v < 0 and geglu:
                continue
            if v == 10 and (K != 320 or mode != 0):
                continue
            if v in (12, 13, 14) and mode not in (0, 1):
                continue
            try:
                outs[v] = call(v, None).clone()
                ok_variants.append(v)
            except RuntimeError as e:
                print("# %s variant %d: %s" % (name, v, e), flush=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        out = torch.empty_like(outs[ok_variants[0]])
        for _ in range(a.rounds):
            for v in ok_variants:
                call(v, out)
                e0.record()
                for _ in range(a.iters):
                    call(v, out)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(1e3 * e0.elapsed_time(e1) / a.iters)
        flop = 2.0 * M * N * K
        row = {"shape": name, "M": M, "N": N, "K": K, "mode": mode}
        base = outs.get(1)
        for v in ok_variants:
            us = statistics.median(times[v])
            row["v%d_us" % v] = round(us, 1)   # "v-1" = round-2 split-K
            row["v%d_TF" % v] = round(flop / us / 1e6, 0)
            if base is not None and v != 1:
                d = (outs[v].float() - base.float()).abs().max().item()
                row["v%d_maxdiff_vs_gemm3" % v] = round(d, 5)
        if base is not None:
            row["ref_absmax"] = round(base.float().abs().max().item(), 3)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
