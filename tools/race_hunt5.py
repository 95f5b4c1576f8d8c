import sys, torch
sys.path.insert(0, ".")
from motionclone_amd import lib, ops
dev = torch.device("cuda:0"); lib.load()
g = torch.Generator(device=dev).manual_seed(0)
def r(*shape, s=1.0): return (torch.randn(*shape, device=dev, generator=g) * s).half()
qb = r(16 * 4096, 960, s=0.5); dob = r(16 * 4096, 320)
nq = r(16 * 4096, 960, s=0.5); nq1 = r(16 * 1024, 1920, s=0.5); kv = r(2 * 77, 640, s=0.5)
a6, w6 = r(65536, 320, s=0.5), r(320, 320, s=0.05)
def tbwd():
    d = torch.empty_like(qb)
    ops.tattn_bwd(qb[:, :320], qb[:, 320:640], qb[:, 640:], dob, d[:, :320], d[:, 320:640], d[:, 640:], 1, 16, 4096, 8, 40)
    return d
def tfwd(): return ops.tattn_fwd(qb[:, :320], qb[:, 320:640], qb[:, 640:], 1, 16, 4096, 8, 40)
ns = torch.cuda.Stream()
noises = {"attn l0": lambda: ops.attn_fwd(nq[:, :320], nq[:, 320:640], nq[:, 640:], 4096, 4096, 8, 40, 16),
          "attn l1 d80": lambda: ops.attn_fwd(nq1[:, :640], nq1[:, 640:1280], nq1[:, 1280:], 1024, 1024, 8, 80, 16),
          "attn cross": lambda: ops.attn_fwd(nq[:, :320], kv[:, :320], kv[:, 320:], 4096, 77, 8, 40, 16, kv_bdiv=16),
          "tattn_fwd": lambda: ops.tattn_fwd(nq[:, :320], nq[:, 320:640], nq[:, 640:], 1, 16, 4096, 8, 40),
          "gemm2 64-tile": lambda: ops.gemm(a6, w6, tile=64),
          "gemm2 128-tile": lambda: ops.gemm(a6, w6, tile=128),
          "geglu": lambda: ops.geglu_fwd(nq1[:, :1280].contiguous()),
          "gn apply": lambda: ops.gn_apply(a6, None, ops.gn_stats(a6, None, 16, 4096, 1e-5), torch.ones(320, device=dev), torch.zeros(320, device=dev), True, 16, 4096)}
for vname, victim in (("tattn_bwd", tbwd), ("tattn_fwd", tfwd)):
    ref = victim().clone(); torch.cuda.synchronize()
    for name, fn in noises.items():
        bad = 0
        for rep in range(3):
            with torch.cuda.stream(ns):
                for _ in range(12): fn()
            o = victim(); torch.cuda.synchronize()
            bad += int(not torch.equal(o, ref))
        print("victim %-10s noise %-14s: %d/3 differ" % (vname, name, bad), flush=True)
