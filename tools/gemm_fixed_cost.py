"""What the per-tile fixed cost of the 256x320 GEMM is made of: MC_GEMM_DEBUG=1 (no output stores), 32 (streaming stores)
against the shipped kernel, on the shapes where it weighs most (K = 640 / 1280 Linear layers, one wave of tiles)."""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _toolslib  # noqa: E401,E702,F401  (tools build of the library: MC_* switches / debug hooks)
import os, sys, torch
sys.path.insert(0, ".")
from motionclone_amd import ops
from tools.gemm_sweep_util import timeit, r
for name, M, N, K, rs in [("l1 to_out+R", 32768, 640, 640, True), ("l1 proj", 32768, 640, 640, False), ("l1 qkv", 32768, 1920, 640, False),
                          ("l0 ff2+R", 131072, 320, 1280, True), ("l2 ff1", 8192, 10240, 1280, False), ("l1 ff2+R", 32768, 640, 2560, True)]:
    x = r(M, K); w = r(N, K, s=0.02); res = r(M, N) if rs else None
    b = torch.randn(1, N, device=x.device)
    ms = timeit(lambda: ops.gemm(x, w, residual=res, bias=b))
    print("MC_GEMM_DEBUG=%s %-12s M=%6d N=%5d K=%4d: %6.1f us  %5.0f TF" % (os.environ.get("MC_GEMM_DEBUG", "0"), name, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9), flush=True)
