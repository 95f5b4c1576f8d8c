"""3x3 conv (implicit GEMM) timings at the config-2 shapes, B = 2: level 0 (64x64), level 1 (32x32), level 2 (16x16)"""
import sys, torch
sys.path.insert(0, ".")
from motionclone_amd import ops
from tools.gemm_sweep_util import timeit, r
F = 32
for name, H, Cin, Cout in [("l0 320->320", 64, 320, 320), ("l0 640->320", 64, 640, 320), ("l1 640->640", 32, 640, 640), ("l1 1280->640", 32, 1280, 640),
                           ("l2 1280->1280", 16, 1280, 1280), ("l2 2560->1280", 16, 2560, 1280)]:
    x = r(F * H * H, Cin); w = r(Cout, 9 * Cin, s=0.02)
    ms = timeit(lambda: ops.gemm(x, w, mode=ops.CONV_S1, geom=(H, H, H, H), m_out=F * H * H))
    print("%-14s M=%6d N=%4d K=%5d: %7.1f us  %5.0f TF" % (name, F * H * H, Cout, 9 * Cin, ms * 1e3, 2.0 * F * H * H * Cout * 9 * Cin / ms / 1e9), flush=True)
