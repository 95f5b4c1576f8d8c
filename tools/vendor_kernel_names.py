"""Which kernels the vendor GEMM runs on the three wide-N, short-K shapes where it beats gemm5 (profiles/r05_vendor_anchor.md):
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -- python tools/vendor_kernel_names.py
then the kernel names / grid / workgroup / LDS / register columns of the trace say what tile and how many workgroups per CU."""
import torch

dev = torch.device("cuda:0")
for M, N, K in ((32768, 5120, 640), (8192, 3840, 1280), (8192, 10240, 1280), (32768, 640, 11520)):
    x = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) * 0.02).half()
    o = torch.empty(M, N, device=dev, dtype=torch.float16)
    for _ in range(3):
        torch.matmul(x, w.t(), out=o)
    torch.cuda.synchronize()
