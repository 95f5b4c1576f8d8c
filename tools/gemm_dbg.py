import sys, torch
sys.path.insert(0, ".")
from motionclone_amd import ops
from tools.gemm_sweep_util import timeit, r
for name, M, N, K, gg in [("qkv_l0", 131072, 960, 320, False), ("ff2_l0", 131072, 320, 1280, False)]:
    x = r(M, K); w = r(N, K, s=0.02)
    ms = timeit(lambda: ops.gemm(x, w, geglu=gg, cfg=1))
    print(name, "%.0f us" % (ms * 1000), flush=True)
F, H, Cin, Cout = 32, 64, 320, 320
x = r(F * H * H, Cin); w = r(Cout, 9 * Cin, s=0.02)
ms = timeit(lambda: ops.gemm(x, w, mode=ops.CONV_S1, geom=(H, H, H, H), m_out=F * H * H, cfg=1))
print("conv_l0", "%.0f us" % (ms * 1000), flush=True)
