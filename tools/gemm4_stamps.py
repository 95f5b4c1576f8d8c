import json, sys, torch
sys.path.insert(0, ".")
from motionclone_amd import ops, lib
dev = torch.device("cuda:0")
M, N = 131072, int(sys.argv[1]) if len(sys.argv) > 1 else 2560
a = (torch.randn(M, 320, device=dev) * 0.5).half(); w = (torch.randn(N, 320, device=dev) * 0.05).half(); b = torch.randn(1, N, device=dev)
nwg = 512
buf = torch.zeros(nwg * 4 * 32 * 6, dtype=torch.int64, device=dev)
L = lib.load()
for _ in range(3): ops.gemm(a, w, bias=b, cfg=10, nsplit=1)
L.mc_gemm_debug_buffer(buf.data_ptr()); L.mc_gemm_debug(16)
ops.gemm(a, w, bias=b, cfg=10, nsplit=1)
torch.cuda.synchronize(); L.mc_gemm_debug(0)
t = buf.view(nwg, 4, 32, 6).cpu().double()
names = ["compute", "vmwait", "barrier", "epilogue_to_regs", "stores_issue", "loop_back"]
d = torch.stack([t[..., 1] - t[..., 0], t[..., 2] - t[..., 1], t[..., 3] - t[..., 2], t[..., 4] - t[..., 3], t[..., 5] - t[..., 4]], -1)  # [wg, wave, chunk, 5]
period = t[:, :, 1:, 0] - t[:, :, :-1, 0]
sel = slice(8, 30)
print("chunk period cycles (median over wgs/waves, chunks 8..30):", period[:, :, sel].median().item(), "mean", period[:, :, sel].mean().item())
for i, n in enumerate(names[:5]):
    x = d[:, :, sel, i]
    print("%-18s median %8.0f  mean %8.0f  p90 %8.0f" % (n, x.median().item(), x.mean().item(), x.flatten().kthvalue(int(0.9 * x.numel())).values.item()))
print("one wave timeline (wg 0 wave 0), chunks 8..14:", d[0, 0, 8:14].tolist())
print("start skew across WGs (first stamp) cycles: min %.0f max %.0f" % ((t[:, 0, 0, 0] - t[:, 0, 0, 0].min()).min().item(), (t[:, 0, 0, 0] - t[:, 0, 0, 0].min()).max().item()))
