"""Temporal attention forward / backward at the shapes of config 2 (CFG batch 2 x 16 frames) on cuda:0 -> JSON lines.
Run once per setting of MC_TATTN_VEC (read once per process)."""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _toolslib  # noqa: E401,E702,F401  (tools build of the library: MC_* switches / debug hooks)
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from motionclone_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=5):
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


tag = os.environ.get("MC_TATTN_VEC", "default")
F = 16
for B in (1, 2):
    for (name, HW, d) in [("l0", 4096, 40), ("l1", 1024, 80), ("l2", 256, 160), ("l3", 64, 160)]:
        C = 8 * d
        qkv = (torch.randn(B * F * HW, 3 * C, device=dev) * 0.5).half()
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        out = torch.empty(B * F * HW, C, device=dev, dtype=torch.float16)
        ms = timeit(lambda: ops.tattn_fwd(q, k, v, B, F, HW, 8, d, out=out))
        nbytes = 8.0 * B * F * HW * C
        do = torch.randn(B * F * HW, C, device=dev).half()
        dqkv = torch.empty_like(qkv)
        msb = timeit(lambda: ops.tattn_bwd(q, k, v, do, dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:], B, F, HW, 8, d))
        print(json.dumps(dict(vec=tag, level=name, B=B, fwd_us=round(1e3 * ms, 2), fwd_gbps=round(nbytes / ms / 1e6, 1),
                              bwd_us=round(1e3 * msb, 2), bwd_gbps=round(14.0 / 8.0 * nbytes / msb / 1e6, 1))), flush=True)
