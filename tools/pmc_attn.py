"""PMC workload: a few launches of the level-0 spatial self-attention (16 frames x 8 heads x 4096 tokens, d = 40)."""
import sys
import torch
sys.path.insert(0, ".")
from motionclone_amd import ops
dev = torch.device("cuda:0")
F, N, d = 16, 4096, 40
C = 8 * d
qkv = (torch.randn(F * N, 3 * C, device=dev) * 0.5).half()
q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
for _ in range(3):
    o, lse = ops.attn_fwd(q, k, v, N, N, 8, d, F)
if len(sys.argv) > 1 and sys.argv[1] == "bwd":
    do = (torch.randn(F * N, C, device=dev)).half()
    for _ in range(2):
        ops.attn_bwd(q, k, v, o, do, lse, N, N, 8, d, F)
torch.cuda.synchronize()
print("ok")
