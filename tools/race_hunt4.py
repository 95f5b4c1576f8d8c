import sys, torch
sys.path.insert(0, ".")
from motionclone_amd import lib, ops
dev = torch.device("cuda:0"); lib.load()
g = torch.Generator(device=dev).manual_seed(0)
def r(*shape, s=1.0): return (torch.randn(*shape, device=dev, generator=g) * s).half()
qb = r(16 * 4096, 960, s=0.5); dob = r(16 * 4096, 320)
na, nw = r(65536, 1280, s=0.5), r(1280, 1280, s=0.03)
nq = r(16 * 4096, 960, s=0.5)
ones, zeros = torch.ones(1280, device=dev), torch.zeros(1280, device=dev)
def tbwd():
    d = torch.empty_like(qb)
    ops.tattn_bwd(qb[:, :320], qb[:, 320:640], qb[:, 640:], dob, d[:, :320], d[:, 320:640], d[:, 640:], 1, 16, 4096, 8, 40)
    return d
ref = tbwd().clone(); torch.cuda.synchronize()
ns = torch.cuda.Stream()
noises = {"gemm": lambda: ops.gemm(na, nw), "attn": lambda: ops.attn_fwd(nq[:, :320], nq[:, 320:640], nq[:, 640:], 4096, 4096, 8, 40, 16),
          "ln": lambda: ops.layernorm_fwd(na, ones, zeros)}
for name, fn in noises.items():
    bad = 0; info = ""
    for rep in range(4):
        with torch.cuda.stream(ns):
            for _ in range(12): fn()
        o = tbwd(); torch.cuda.synchronize()
        if not torch.equal(o, ref):
            bad += 1
            dd = (o.float() - ref.float()).abs()
            info = "cols with diff: dq %d dk %d dv %d; rows %d; max %.4g" % (int((dd[:, :320] > 0).sum()), int((dd[:, 320:640] > 0).sum()), int((dd[:, 640:] > 0).sum()), int((dd.amax(1) > 0).sum()), float(dd.max()))
    print("noise %-5s: %d/4 differ  %s" % (name, bad, info), flush=True)
# same-stream interleave (no concurrency): noise then victim on the default stream
for name, fn in noises.items():
    for _ in range(12): fn()
    o = tbwd(); torch.cuda.synchronize()
    print("same-stream after %-5s identical: %s" % (name, bool(torch.equal(o, ref))))
