import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _toolslib  # noqa: E401,E702,F401  (tools build of the library: MC_* switches / debug hooks)
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
o_ref = None
print(os.environ.get("MC_HIP_LIB", "default"), "QT", os.environ.get("MC_ATTN_QT", "auto"), "attn_fwd level 0, B=2 (32 frames): %.3f ms" % timeit(lambda: ops.attn_fwd(q, k, v, N, N, 8, d, F)))
o, lse = ops.attn_fwd(q, k, v, N, N, 8, d, F)
ref = torch.nn.functional.scaled_dot_product_attention(*[t[:4 * N].view(4, N, 8, d).transpose(1, 2).float() for t in (q, k, v)])
print("  max |o - sdpa| on 4 frames: %.3e" % (o[:4 * N].view(4, N, 8, d).transpose(1, 2).float() - ref).abs().max().item())
do = (torch.randn(F * N, 8 * d, device=dev) * 0.5).half()
F1 = 16   # the taped half runs at B = 1
dqkv = torch.zeros(F1 * N, 3 * 8 * d, device=dev, dtype=torch.float16)
C = 8 * d
def bwd():
    ops.attn_bwd(q[:F1 * N], k[:F1 * N], v[:F1 * N], o[:F1 * N], do[:F1 * N], lse[:F1], N, N, 8, d, F1,
                 dq=dqkv[:, :C], dk=dqkv[:, C:2 * C], dv=dqkv[:, 2 * C:])
try:
    print("  BQT", os.environ.get("MC_ATTN_BQT", "auto"), "attn_bwd level 0, B=1 (16 frames): %.3f ms" % timeit(bwd, 5), " checksum %.6f" % dqkv.float().abs().mean().item())
except Exception as e:
    print("  bwd failed:", e)
