import sys, torch
sys.path.insert(0, ".")
from motionclone_amd import lib, ops
dev = torch.device("cuda:0"); lib.load()
g = torch.Generator(device=dev).manual_seed(0)
def r(*shape, s=1.0): return (torch.randn(*shape, device=dev, generator=g) * s).half()
qb = r(16 * 4096, 960, s=0.5); dob = r(16 * 4096, 320)
nq = r(16 * 4096, 960, s=0.5)
ridx = torch.randint(0, 16, (4096, 8, 16, 1), device=dev, generator=g).to(torch.uint8)
rval = torch.rand((4096, 8, 16, 1), device=dev, generator=g) * 0.5
def v_full():
    d = torch.empty_like(qb)
    ops.tattn_bwd(qb[:, :320], qb[:, 320:640], qb[:, 640:], dob, d[:, :320], d[:, 320:640], d[:, 640:], 1, 16, 4096, 8, 40)
    return d
def v_seed():
    d = torch.empty_like(qb)
    ops.tattn_bwd(qb[:, :320], qb[:, 320:640], qb[:, 640:], None, d[:, :320], d[:, 320:640], d[:, 640:], 1, 16, 4096, 8, 40, ref_idx=ridx, ref_val=rval, seed_coef=3.0)
    return d
def v_both():
    d = torch.empty_like(qb)
    ops.tattn_bwd(qb[:, :320], qb[:, 320:640], qb[:, 640:], dob, d[:, :320], d[:, 320:640], d[:, 640:], 1, 16, 4096, 8, 40, ref_idx=ridx, ref_val=rval, seed_coef=3.0)
    return d
ns = torch.cuda.Stream()
def noise():
    with torch.cuda.stream(ns):
        for _ in range(12): ops.attn_fwd(nq[:, :320], nq[:, 320:640], nq[:, 640:], 4096, 4096, 8, 40, 16)
for name, v in (("dO only", v_full), ("seed only", v_seed), ("dO + seed", v_both)):
    ref = v().clone(); torch.cuda.synchronize()
    bad = 0
    for _ in range(4):
        noise(); o = v(); torch.cuda.synchronize(); bad += int(not torch.equal(o, ref))
    print("%-10s under attn noise: %d/4 differ" % (name, bad), flush=True)
