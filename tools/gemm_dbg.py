import sys, torch
sys.path.insert(0, ".")
from motionclone_amd import ops
from tools.gemm_sweep_util import timeit, r
for name, M, N, K, gg in [("qkv_l0", 131072, 960, 320, False), ("ff1_l0", 131072, 2560, 320, True), ("ff2_l0", 131072, 320, 1280, False), ("o_l0", 131072, 320, 320, False)]:
    x = r(M, K); w = r(N, K, s=0.02)
    ms = timeit(lambda: ops.gemm(x, w, geglu=gg, cfg=1))
    print(name, "%.0f us" % (ms * 1000), flush=True)
