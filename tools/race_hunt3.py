import sys, torch
sys.path.insert(0, ".")
from motionclone_amd import lib, ops
dev = torch.device("cuda:0"); lib.load()
g = torch.Generator(device=dev).manual_seed(0)
def r(*shape, s=1.0): return (torch.randn(*shape, device=dev, generator=g) * s).half()
# guard tensors interleaved with the noise operands: any out-of-bounds write lands in one of them
guards = []
def guard():
    t = torch.full((1 << 20,), 3.0, device=dev, dtype=torch.float16); guards.append(t); return t
guard(); na = r(65536, 1280, s=0.5); guard(); nw = r(1280, 1280, s=0.03); guard()
nq = r(16 * 4096, 960, s=0.5); guard()
ones, zeros = torch.ones(1280, device=dev), torch.zeros(1280, device=dev); guard()
def check(label):
    torch.cuda.synchronize()
    bad = [i for i, t in enumerate(guards) if not bool((t == 3.0).all())]
    print("%-28s guards corrupted: %s" % (label, bad), flush=True)
    for t in guards: t.fill_(3.0)
for label, fn in (("gemm 65536x1280x1280", lambda: ops.gemm(na, nw)),
                  ("attn_fwd l0 nbatch 16", lambda: ops.attn_fwd(nq[:, :320], nq[:, 320:640], nq[:, 640:], 4096, 4096, 8, 40, 16)),
                  ("layernorm C=1280", lambda: ops.layernorm_fwd(na, ones, zeros)),
                  ("layernorm C=320", lambda: ops.layernorm_fwd(nq[:, :320].contiguous(), ones[:320].contiguous(), zeros[:320].contiguous())),
                  ("layernorm C=640", lambda: ops.layernorm_fwd(nq[:, :640].contiguous(), ones[:640].contiguous(), zeros[:640].contiguous()))):
    outs = []
    for _ in range(3):
        guard_pre = torch.full((1 << 18,), 3.0, device=dev, dtype=torch.float16)
        o = fn()
        guard_post = torch.full((1 << 18,), 3.0, device=dev, dtype=torch.float16)
        outs.append((guard_pre, o, guard_post))
    torch.cuda.synchronize()
    ok = all(bool((a == 3.0).all()) and bool((c == 3.0).all()) for a, _, c in outs)
    print("%-28s output neighbours intact: %s" % (label, ok))
    check(label)
# inputs of the LN call: na itself must be unchanged by LN (it is an input)
na0 = na.clone(); ops.layernorm_fwd(na, ones, zeros); torch.cuda.synchronize(); print("LN input unchanged:", bool(torch.equal(na, na0)))
nq0 = nq.clone(); ops.attn_fwd(nq[:, :320], nq[:, 320:640], nq[:, 640:], 4096, 4096, 8, 40, 16); torch.cuda.synchronize(); print("attn input unchanged:", bool(torch.equal(nq, nq0)))
