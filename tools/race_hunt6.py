import sys, torch
sys.path.insert(0, ".")
from motionclone_amd import lib, ops
dev = torch.device("cuda:0"); lib.load()
g = torch.Generator(device=dev).manual_seed(0)
def r(*shape, s=1.0): return (torch.randn(*shape, device=dev, generator=g) * s).half()
qb = r(16 * 4096, 960, s=0.5); dob = r(16 * 4096, 320)
nq = r(16 * 4096, 960, s=0.5)
def tbwd(v=None, do=None):
    d = torch.empty_like(qb)
    ops.tattn_bwd(qb[:, :320], qb[:, 320:640], qb[:, 640:] if v is None else v, dob if do is None else do, d[:, :320], d[:, 320:640], d[:, 640:], 1, 16, 4096, 8, 40)
    return d
ns = torch.cuda.Stream()
def noise():
    with torch.cuda.stream(ns):
        for _ in range(12): ops.attn_fwd(nq[:, :320], nq[:, 320:640], nq[:, 640:], 4096, 4096, 8, 40, 16)
ref = tbwd().clone(); torch.cuda.synchronize()
noise(); o = tbwd(); torch.cuda.synchronize()
dd = (o.float() - ref.float()).abs()
rows = (dd.amax(1) > 0).nonzero().flatten()
print("n rows differ", rows.numel(), "first rows", rows[:24].tolist())
if rows.numel():
    r0 = int(rows[0]); f, p = r0 // 4096, r0 % 4096
    cols = (dd[r0] > 0).nonzero().flatten().tolist()
    print("row", r0, "frame", f, "pixel", p, "cols differing", cols[:6], "...", len(cols))
    h = cols[0] // 40 if cols[0] < 320 else (cols[0] - 320) // 40
    sl = slice(40 * h, 40 * h + 8)
    print("dq ref ", ref[r0, sl].float().tolist()); print("dq got ", o[r0, sl].float().tolist())
    print("ratio", (o[r0, sl].float() / ref[r0, sl].float()).tolist())
    px_rows = [ff * 4096 + p for ff in range(16)]
    print("frames of this pixel differing:", [int(dd[x].max() > 0) for x in px_rows])
    # pixels affected: contiguous range?
    pix = sorted(set((rows % 4096).tolist())); print("pixels affected", len(pix), pix[:40])
# variants
#vc = qb[:, 640:].contiguous()
#refv = tbwd(v=vc).clone(); torch.cuda.synchronize()
#print("contiguous v solo == ref:", bool(torch.equal(refv, ref)))
import sys; sys.exit(0)
for _ in range(3):
    noise(); o2 = tbwd(v=vc); torch.cuda.synchronize(); bad += int(not torch.equal(o2, ref))
print("contiguous v under attn noise differ:", bad, "/3")
