import os, sys, torch
sys.path.insert(0, ".")
from motionclone_amd import lib, ops
dev = torch.device("cuda:0"); lib.load()
F, N, d = 32, 4096, 40
q, k, v = [(torch.randn(F * N, 8 * d, device=dev) * 0.5).half() for _ in range(3)]
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
print(os.environ.get("MC_HIP_LIB", "default"), "attn_fwd level 0, B=2 (32 frames): %.3f ms" % timeit(lambda: ops.attn_fwd(q, k, v, N, N, 8, d, F)))
