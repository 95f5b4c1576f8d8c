"""GroupNorm forward at the config-2 shapes: stats + apply (three launches) vs the fused entry (two launches) -> JSON lines."""
import json
import sys

import torch

sys.path.insert(0, ".")
from motionclone_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for frames in (16, 32):
    for (hw, C) in [(4096, 320), (1024, 640), (256, 1280), (64, 1280), (4096, 640)]:
        x = torch.randn(frames * hw, C, device=dev).half()
        g = torch.ones(C, device=dev)
        b = torch.zeros(C, device=dev)
        out = torch.empty_like(x)
        t3 = timeit(lambda: ops.gn_apply(x, None, ops.gn_stats(x, None, frames, hw, 1e-5), g, b, True, frames, hw, out=out))
        t2 = timeit(lambda: ops.gn_fwd(x, None, g, b, True, frames, hw, 1e-5, out=out))
        dz = torch.randn_like(x)
        _, st = ops.gn_fwd(x, None, g, b, True, frames, hw, 1e-5, out=out)
        tb = timeit(lambda: ops.gn_bwd(x, None, dz, st, g, b, True, frames, hw, out=out))
        print(json.dumps(dict(frames=frames, hw=hw, C=C, stats_apply_us=round(t3, 1), fused_us=round(t2, 1), bwd_us=round(tb, 1),
                              fused_gbps=round(6.0 * frames * hw * C / t2 / 1e3, 1))), flush=True)
