"""Small workload for PMC runs: a few launches of the dominant GEMM/conv kernels at config-2 shapes."""
import sys
import torch
sys.path.insert(0, ".")
which = sys.argv[1] if len(sys.argv) > 1 else "mc"
dev = torch.device("cuda:0")
if which == "torch":
    a = torch.randn(4096, 4096, device=dev).half()
    for _ in range(3):
        b = a @ a
    torch.cuda.synchronize()
    print("torch ok")
    sys.exit(0)
from motionclone_amd import ops
F = 16
def r(*s, sc=1.0):
    return (torch.randn(*s, device=dev) * sc).half()
x = r(F * 64 * 64, 320); w = r(320, 9 * 320, sc=0.02)
for _ in range(3):
    ops.gemm(x, w, mode=ops.CONV_S1, geom=(64, 64, 64, 64), m_out=F * 64 * 64)      # gemm3<CONV_S1,256,320>
xq = r(65536, 320); wq = r(960, 320, sc=0.02); wf = r(2560, 320, sc=0.02); w2 = r(320, 1280, sc=0.02); x2 = r(65536, 1280)
res = r(65536, 320)
for _ in range(3):
    ops.gemm(xq, wq)                       # qkv
    ops.gemm(xq, wf, geglu=True)           # ff1 + geglu
    ops.gemm(x2, w2, residual=res)         # ff2
torch.cuda.synchronize()
print("mc ok")
