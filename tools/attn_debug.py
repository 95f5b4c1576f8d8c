import sys, torch
sys.path.insert(0, '.')
from motionclone_amd import ops
torch.manual_seed(0)
dev = torch.device('cuda:0')
def run(d, Nq, heads, nb):
    C = heads*d
    qkv = (torch.randn(nb*Nq, 3*C)*0.7).half().to(dev)
    q,k,v = qkv[:, :C], qkv[:, C:2*C], qkv[:, 2*C:]
    o, lse = ops.attn_fwd(q,k,v,Nq,Nq,heads,d,nb)
    def H(t): return t.float().reshape(nb,Nq,heads,d).permute(0,2,1,3)
    S = (H(q) @ H(k).transpose(-1,-2))*d**-0.5
    ref = S.softmax(-1) @ H(v)
    err = (H(o)-ref).abs()
    bad = (err > 1e-2)
    print("d", d, "Nq", Nq, "max err %.4f bad frac %.4f" % (err.max().item(), bad.float().mean().item()))
    if bad.any():
        idx = bad.nonzero()
        print(" bad batches", idx[:,0].unique().tolist(), "heads", idx[:,1].unique().tolist())
        rows = idx[:,2].unique()
        print(" bad q rows: n=%d min %d max %d  first %s" % (len(rows), rows.min(), rows.max(), rows[:24].tolist()))
        print(" bad d cols", idx[:,3].unique().tolist()[:40])
        lerr = (lse.cpu() - torch.logsumexp(S, -1).cpu()).abs()
        print(" lse max err", lerr.max().item())
for d, Nq in [(80, 579), (80, 512), (80, 128), (160, 300), (40, 579)]:
    run(d, Nq, 4, 3)
