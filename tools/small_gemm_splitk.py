"""The small Linear layers of the 8x8 / 16x16 levels (guided step, batch 1): library plan vs forced split-K -> JSON lines."""
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


for (M, N, K, res) in [(1024, 1280, 1280, False), (2048, 1280, 1280, True), (4096, 1280, 1280, False), (1024, 3840, 1280, False),
                       (2048, 3840, 1280, False), (1024, 1280, 5120, True), (4096, 1280, 5120, True)]:
    x = (torch.randn(M, K, device=dev)).half()
    w = (torch.randn(N, K, device=dev) * 0.02).half()
    r = torch.randn(M, N, device=dev).half() if res else None
    row = dict(M=M, N=N, K=K, residual=res, plan_us=round(timeit(lambda: ops.gemm(x, w, residual=r)), 1))
    from motionclone_amd import lib
    row["plan_kernel"] = lib.load().mc_gemm_last_kernel()
    for sp in (2, 4, 8):
        if K // sp % 64 == 0:
            row["split%d_us" % sp] = round(timeit(lambda: ops.gemm(x, w, residual=r, splits=sp)), 1)
            row["split%d_kernel" % sp] = lib.load().mc_gemm_last_kernel()
    print(json.dumps(row), flush=True)
