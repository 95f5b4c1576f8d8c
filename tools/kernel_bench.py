"""Micro-benchmark of the hot kernel shapes of config 2 (16f x 512^2) on cuda:0 -> JSON lines."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from motionclone_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def r(*shape, s=1.0):
    return (torch.randn(*shape, device=dev) * s).half()


out = []
F = 16
for (name, H, Cin, Cout) in [("conv_l0", 64, 320, 320), ("conv_l1", 32, 640, 640), ("conv_l2", 16, 1280, 1280),
                             ("conv_l3", 8, 1280, 1280), ("conv_up3cat", 64, 640, 320), ("conv_up1cat", 16, 2560, 1280)]:
    x = r(F * H * H, Cin)
    w = r(Cout, 9 * Cin, s=0.02)
    for tile in (128, 64):
        for v in ("v1", "v2", "v3"):
            ms = timeit(lambda: ops.gemm(x, w, mode=ops.CONV_S1, geom=(H, H, H, H), m_out=F * H * H, tile=tile,
                                         v1=v == "v1", deep=v == "v3"))
            fl = 2.0 * F * H * H * Cout * 9 * Cin
            out.append(dict(k=name, tile=tile, v=v, ms=ms, tflops=fl / ms / 1e9))
    for cfg in (1, 2, 3, 4, 5):
        ms = timeit(lambda: ops.gemm(x, w, mode=ops.CONV_S1, geom=(H, H, H, H), m_out=F * H * H, cfg=cfg))
        out.append(dict(k=name, tile=0, v="g3c%d" % cfg, ms=ms, tflops=2.0 * F * H * H * Cout * 9 * Cin / ms / 1e9))
for (name, M, N, K) in [("lin_o_l0", 65536, 320, 320), ("lin_qkv_l0", 65536, 960, 320), ("lin_ff1_l0", 65536, 2560, 320), ("lin_ff2_l0", 65536, 320, 1280),
                        ("lin_ff1_l1", 16384, 5120, 640), ("lin_ff1_l2", 4096, 10240, 1280), ("lin_o_l2", 4096, 1280, 1280),
                        ("lin_l3", 1024, 1280, 1280)]:
    x = r(M, K)
    w = r(N, K, s=0.02)
    for tile in (128, 64):
        for v in ("v1", "v2", "v3"):
            ms = timeit(lambda: ops.gemm(x, w, tile=tile, v1=v == "v1", deep=v == "v3"))
            out.append(dict(k=name, tile=tile, v=v, ms=ms, tflops=2.0 * M * N * K / ms / 1e9))
    for cfg in (1, 2, 3, 4, 5):
        ms = timeit(lambda: ops.gemm(x, w, cfg=cfg))
        out.append(dict(k=name, tile=0, v="g3c%d" % cfg, ms=ms, tflops=2.0 * M * N * K / ms / 1e9))
    if N % 16 == 0:
        ms = timeit(lambda: ops.gemm(x, w, tile=128, geglu=True))
        out.append(dict(k=name + "+geglu", tile=128, v="v2", ms=ms, tflops=2.0 * M * N * K / ms / 1e9))
for (name, N, d) in [("attn_l0", 4096, 40), ("attn_l1", 1024, 80), ("attn_l2", 256, 160)]:
    C = 8 * d
    qkv = r(F * N, 3 * C, s=0.5)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    ms = timeit(lambda: ops.attn_fwd(q, k, v, N, N, 8, d, F))
    fl = 4.0 * F * 8 * N * N * d
    out.append(dict(k=name + "_fwd", ms=ms, tflops=fl / ms / 1e9))
    o, lse = ops.attn_fwd(q, k, v, N, N, 8, d, F)
    do = r(F * N, C)
    ms = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, N, N, 8, d, F), iters=5)
    out.append(dict(k=name + "_bwd", ms=ms, tflops=2.5 * fl / ms / 1e9))
for (name, HW, d) in [("tattn_l0", 4096, 40), ("tattn_l2", 256, 160)]:
    C = 8 * d
    qkv = r(F * HW, 3 * C, s=0.5)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    ms = timeit(lambda: ops.tattn_fwd(q, k, v, 1, F, HW, 8, d))
    out.append(dict(k=name + "_fwd", ms=ms, gbps=qkv.numel() * 2 * 4 / 3 / ms / 1e6))
    dq = torch.empty_like(qkv)
    do = r(F * HW, C)
    ms = timeit(lambda: ops.tattn_bwd(q, k, v, do, dq[:, :C], dq[:, C:2 * C], dq[:, 2 * C:], 1, F, HW, 8, d))
    out.append(dict(k=name + "_bwd", ms=ms))
for (name, H, C) in [("gn_l0", 64, 320), ("gn_l0cat", 64, 640), ("gn_l2", 16, 1280)]:
    x = r(F * H * H, C)
    g = torch.ones(C, device=dev)
    b = torch.zeros(C, device=dev)
    ms1 = timeit(lambda: ops.gn_stats(x, None, F, H * H, 1e-5))
    st = ops.gn_stats(x, None, F, H * H, 1e-5)
    ms2 = timeit(lambda: ops.gn_apply(x, None, st, g, b, True, F, H * H))
    out.append(dict(k=name, stats_ms=ms1, apply_ms=ms2, stats_gbps=x.numel() * 2 / ms1 / 1e6,
                    apply_gbps=x.numel() * 4 / ms2 / 1e6))
x = r(65536, 320)
g = torch.ones(320, device=dev)
b = torch.zeros(320, device=dev)
ms = timeit(lambda: ops.layernorm_fwd(x, g, b))
out.append(dict(k="ln_l0", ms=ms, gbps=x.numel() * 4 / ms / 1e6))
x = r(65536, 2560)
ms = timeit(lambda: ops.geglu_fwd(x))
out.append(dict(k="geglu_l0", ms=ms, gbps=x.numel() * 3 / ms / 1e6))
for o in out:
    print(json.dumps(o))
