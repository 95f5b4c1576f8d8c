"""CPU cost of one wrapper call -> kernel launch (tiny problems, no sync inside the loop)."""
import sys, time, torch
sys.path.insert(0, ".")
from motionclone_amd import ops
dev = torch.device("cuda:0")
a = torch.randn(64, 64, device=dev).half(); w = torch.randn(64, 64, device=dev).half()
x = torch.randn(256, 64, device=dev).half()
g = torch.ones(64, device=dev); b = torch.zeros(64, device=dev)
def bench(name, fn, n=3000):
    for _ in range(100): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-28s %.2f us/call issue, %.2f us/call incl. drain" % (name, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
bench("ops.gemm 64^3", lambda: ops.gemm(a, w))
out = torch.empty(64, 64, device=dev, dtype=torch.float16)
bench("ops.gemm 64^3 (out=)", lambda: ops.gemm(a, w, out=out))
bench("ops.silu", lambda: ops.silu(a))
bench("ops.layernorm_fwd", lambda: ops.layernorm_fwd(x, g, b))
bench("torch.empty", lambda: torch.empty(64, 64, device=dev, dtype=torch.float16))
bench("current_stream", lambda: torch.cuda.current_stream(dev).cuda_stream)
bench("torch a+a", lambda: a + a)
