"""Small workload for PMC runs: a few launches of the dominant GEMM / conv kernels at config-2 shapes (B = 2 forward:
131072 token rows at the 64x64 level).  Prints the launch order so counter rows can be matched to shapes."""
import sys
import torch
sys.path.insert(0, ".")
dev = torch.device("cuda:0")
from motionclone_amd import ops
F = 32
def r(*s, sc=1.0):
    return (torch.randn(*s, device=dev) * sc).half()
x = r(F * 64 * 64, 320); w = r(320, 9 * 320, sc=0.02)
xq = r(131072, 320); wq = r(960, 320, sc=0.02); wf = r(2560, 320, sc=0.02); wo = r(320, 320, sc=0.02)
w2 = r(320, 1280, sc=0.02); x2 = r(131072, 1280); res = r(131072, 320)
x1 = r(32768, 640); wq1 = r(1920, 640, sc=0.02)
torch.cuda.synchronize()
shapes = ["conv3x3 M=131072 N=320 K=2880 (gemm3 CONV_S1 256x320)", "qkv M=131072 N=960 K=320 (gemm4)",
          "ff1+GEGLU M=131072 N=2560 K=320 (gemm4)", "to_out+R M=131072 N=320 K=320 (gemm4)",
          "ff2+R M=131072 N=320 K=1280 (gemm3 DENSE 256x320)", "qkv level 1 M=32768 N=1920 K=640 (gemm3 DENSE 256x320)"]
for _ in range(3):
    ops.gemm(x, w, mode=ops.CONV_S1, geom=(64, 64, 64, 64), m_out=F * 64 * 64)
    ops.gemm(xq, wq)
    ops.gemm(xq, wf, geglu=True)
    ops.gemm(xq, wo, residual=res)
    ops.gemm(x2, w2, residual=res)
    ops.gemm(x1, wq1)
torch.cuda.synchronize()
print("launch order x3:", shapes)
