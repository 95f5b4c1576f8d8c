import json, sys, torch
sys.path.insert(0, ".")
from motionclone_amd import ops, lib
dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, res) in [(131072, 2560, False), (131072, 960, False), (131072, 320, True)]:
    a = (torch.randn(M, 320, device=dev) * 0.5).half(); w = (torch.randn(N, 320, device=dev) * 0.05).half()
    b = torch.randn(1, N, device=dev); r = torch.randn(M, N, device=dev).half() if res else None
    row = dict(M=M, N=N, res=res)
    for name, bits in [("full", 0), ("nostore", 1), ("nomfma", 2), ("noepi", 4), ("noW", 8), ("nomfma_noepi", 6), ("nomfma_noepi_noW", 14), ("nostore_nomfma", 3)]:
        lib.load().mc_gemm_debug(bits)
        row[name] = round(timeit(lambda: ops.gemm(a, w, bias=b, residual=r, cfg=10, nsplit=1)), 1)
    lib.load().mc_gemm_debug(0)
    print(json.dumps(row))
