import sys, torch
sys.path.insert(0, ".")
from motionclone_amd import ops
from tools.gemm_sweep_util import timeit, r
for name, M, N, K, gg, rs in [("qkv_l0", 131072, 960, 320, False, False), ("ff1_l0", 131072, 2560, 320, True, False), ("ff2_l0", 131072, 320, 1280, False, True), ("o_l0", 131072, 320, 320, False, True)]:
    x = r(M, K); w = r(N, K, s=0.02); res = r(M, N) if rs else None
    b = torch.randn(1, N, device=x.device)
    ms = timeit(lambda: ops.gemm(x, w, geglu=gg, residual=res, bias=b if rs else None))
    print(name, "%.0f us" % (ms * 1000), flush=True)
