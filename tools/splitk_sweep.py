"""Which (geometry, K ranges) wins on the mid-size 3x3 convs (levels 1-3 of config 2)."""
import sys, torch
sys.path.insert(0, ".")
from motionclone_amd import ops
from tools.gemm_sweep_util import timeit, r
cases = [("L2 B2", 32, 16, 1280, 1280), ("L2 B1", 16, 16, 1280, 1280), ("L2cat B2", 32, 16, 2560, 1280), ("L1 B1", 16, 32, 640, 640),
         ("L1 B2", 32, 32, 640, 640), ("L1cat B1", 16, 32, 1280, 640), ("L1cat B2", 32, 32, 1920, 640), ("L3 B2", 32, 8, 1280, 1280), ("L0 B1", 16, 64, 320, 320)]
variants = [("auto", {}), ("c1", dict(cfg=1, splits=1)), ("c4", dict(cfg=4, splits=1)), ("c1s2", dict(cfg=1, splits=2)), ("c1s3", dict(cfg=1, splits=3)),
            ("c1s4", dict(cfg=1, splits=4)), ("c4s2", dict(cfg=4, splits=2)), ("c4s4", dict(cfg=4, splits=4))]
for name, F, H, Cin, Cout in cases:
    x = r(F * H * H, Cin); w = r(Cout, 9 * Cin, s=0.02)
    row = ["%-9s" % name]
    for label, kw in variants:
        ms = timeit(lambda: ops.gemm(x, w, mode=ops.CONV_S1, geom=(H, H, H, H), m_out=F * H * H, **kw))
        row.append("%s %.0f" % (label, ms * 1000))
    print(" | ".join(row), flush=True)
