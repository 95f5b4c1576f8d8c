"""Bandwidth of the memory-bound kernels at the B = 2 level-0/1 shapes of config 2 (algorithmic bytes / time)."""
import sys, torch
sys.path.insert(0, ".")
from motionclone_amd import ops
from tools.gemm_sweep_util import timeit, r
dev = torch.device("cuda:0")
for (name, fr, hw, C) in [("L0", 32, 4096, 320), ("L1", 32, 1024, 640), ("L2", 32, 256, 1280), ("L0cat", 32, 4096, 640)]:
    T = fr * hw
    x = r(T, C); g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    ms = timeit(lambda: ops.gn_stats(x, None, fr, hw, 1e-5))
    st = ops.gn_stats(x, None, fr, hw, 1e-5)
    print("gn_stats  %-6s %.1f us  %.0f GB/s" % (name, ms * 1e3, T * C * 2 / ms / 1e6))
    ms = timeit(lambda: ops.gn_apply(x, None, st, g, b, True, fr, hw))
    print("gn_apply  %-6s %.1f us  %.0f GB/s" % (name, ms * 1e3, 2 * T * C * 2 / ms / 1e6))
    ms = timeit(lambda: ops.layernorm_fwd(x, g, b))
    print("ln_fwd    %-6s %.1f us  %.0f GB/s" % (name, ms * 1e3, 2 * T * C * 2 / ms / 1e6))
for (name, B, F, HW, d) in [("tattn L0 B1", 1, 16, 4096, 40), ("tattn L0 B2", 2, 16, 4096, 40), ("tattn L1 B2", 2, 16, 1024, 80), ("tattn L2 B2", 2, 16, 256, 160)]:
    C = 8 * d
    qkv = r(B * F * HW, 3 * C, s=0.5)
    ms = timeit(lambda: ops.tattn_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, F, HW, 8, d))
    print("%-12s %.1f us  %.0f GB/s" % (name, ms * 1e3, qkv.numel() * 2 * 4 / 3 / ms / 1e6))
