"""Sweep of the GEMM kernel geometries on the dense / conv shapes of config 2 -> one line per (shape, variant)."""
import sys
import torch
sys.path.insert(0, ".")
from motionclone_amd import ops


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
    return (torch.randn(*shape, device=torch.device("cuda:0")) * s).half()


dev = torch.device("cuda:0")
shapes = [("o_l0", 131072, 320, 320, False), ("qkv_l0", 131072, 960, 320, False), ("ff1_l0", 131072, 2560, 320, True),
          ("ff1_l0_nogeglu", 131072, 2560, 320, False), ("ff2_l0", 131072, 320, 1280, False),
          ("o_l1", 32768, 640, 640, False), ("qkv_l1", 32768, 1920, 640, False), ("ff1_l1", 32768, 5120, 640, True),
          ("ff2_l1", 32768, 640, 2560, False), ("ff1_l2", 8192, 10240, 1280, True), ("ff2_l2", 8192, 1280, 5120, False)]
for name, M, N, K, gg in shapes:
    x = r(M, K); w = r(N, K, s=0.02); res = r(M, N) if not gg else None
    row = [name]
    for label, kw in [("c1", dict(cfg=1)), ("c6", dict(cfg=6)), ("c7", dict(cfg=7)), ("c8", dict(cfg=8)), ("c9", dict(cfg=9)), ("v2", dict(tile=128))]:
        ms = timeit(lambda: ops.gemm(x, w, residual=res, geglu=gg, **kw))
        row.append("%s %.0fus %.0fTF" % (label, ms * 1000, 2.0 * M * N * K / ms / 1e9))
    print(" | ".join(row), flush=True)
F = 32
for (name, H, Cin, Cout) in [("conv_l0", 64, 320, 320), ("conv_l1", 32, 640, 640), ("conv_l2", 16, 1280, 1280), ("conv_up3cat", 64, 640, 320)]:
    x = r(F * H * H, Cin); w = r(Cout, 9 * Cin, s=0.02)
    row = [name]
    for label, kw in [("c1", dict(cfg=1)), ("c4", dict(cfg=4)), ("c7", dict(cfg=7)), ("c8", dict(cfg=8)), ("c9", dict(cfg=9))]:
        ms = timeit(lambda: ops.gemm(x, w, mode=ops.CONV_S1, geom=(H, H, H, H), m_out=F * H * H, **kw))
        row.append("%s %.0fus %.0fTF" % (label, ms * 1000, 2.0 * F * H * H * Cout * 9 * Cin / ms / 1e9))
    print(" | ".join(row), flush=True)
