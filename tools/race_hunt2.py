import sys, torch
sys.path.insert(0, ".")
from motionclone_amd import lib, ops
dev = torch.device("cuda:0"); lib.load()
g = torch.Generator(device=dev).manual_seed(0)
def r(*shape, s=1.0): return (torch.randn(*shape, device=dev, generator=g) * s).half()
qb = r(16 * 4096, 960, s=0.5); dob = r(16 * 4096, 320)
def tbwd(fill=None, seed=False):
    d = torch.empty_like(qb)
    if fill is not None: d.fill_(fill)
    ops.tattn_bwd(qb[:, :320], qb[:, 320:640], qb[:, 640:], dob, d[:, :320], d[:, 320:640], d[:, 640:], 1, 16, 4096, 8, 40)
    return d
ref = tbwd(0.0).clone(); torch.cuda.synchronize()
for name, fill in (("zero", 0.0), ("nan", float("nan")), ("7", 7.0), ("empty", None)):
    o = tbwd(fill); torch.cuda.synchronize()
    print("prefill", name, "identical:", bool(torch.equal(o, ref)), "isfinite:", bool(torch.isfinite(o.float()).all()), "maxdiff", float((o.float() - ref.float()).abs().nan_to_num(1e9).max()))
# allocator-state change without concurrency
junk = [torch.randn(1 << 24, device=dev) for _ in range(8)]
o = tbwd(None); torch.cuda.synchronize(); print("after junk allocs identical:", bool(torch.equal(o, ref)))
del junk
# concurrency with a pure-compute noise kernel vs memory noise
ns = torch.cuda.Stream()
big = torch.randn(1 << 26, device=dev)
for label, fn in (("mem noise (copy)", lambda: big.clone()), ("gemm noise", lambda: ops.gemm(qb[:, :320].contiguous(), r(320, 320)))):
    bad = 0
    for rep in range(4):
        with torch.cuda.stream(ns):
            for _ in range(20): fn()
        o = tbwd(0.0); torch.cuda.synchronize()
        bad += int(not torch.equal(o, ref))
    print(label, "differ", bad, "/4")
# which output part differs?
with torch.cuda.stream(ns):
    for _ in range(30): big.clone()
o = tbwd(0.0); torch.cuda.synchronize()
for nm, sl in (("dq", slice(0, 320)), ("dk", slice(320, 640)), ("dv", slice(640, 960))):
    dd = (o[:, sl].float() - ref[:, sl].float()).abs()
    print(nm, "max diff", float(dd.max()), "n diff", int((dd > 0).sum()), "rows with diff", int((dd.amax(1) > 0).sum()))
