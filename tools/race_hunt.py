"""Which kernel changes its result when another stream competes for the GPU?  Each candidate runs solo (reference), then
repeatedly while a second stream hammers the chip with unrelated kernels; outputs are compared bitwise."""
import sys, torch
sys.path.insert(0, ".")
from motionclone_amd import lib, ops
dev = torch.device("cuda:0")
lib.load()
g = torch.Generator(device=dev).manual_seed(0)
def r(*shape, s=1.0): return (torch.randn(*shape, device=dev, generator=g) * s).half()
F = 32
cands = {}
# GEMM family
a, w, res = r(131072, 320, s=0.5), r(960, 320, s=0.05), None
cands["gemm4 qkv"] = lambda: ops.gemm(a, w)
w2, r2 = r(320, 320, s=0.05), r(131072, 320)
cands["gemm4 N=320 +R"] = lambda: ops.gemm(a, w2, residual=r2)
a3, w3 = r(32768, 640, s=0.5), r(1920, 640, s=0.05)
cands["gemm3 dense 32768x1920x640"] = lambda: ops.gemm(a3, w3)
a4, w4 = r(8192, 1280, s=0.5), r(1280, 1280, s=0.03)
r4 = r(8192, 1280)
cands["gemm3 cfg4 8192x1280x1280 +R"] = lambda: ops.gemm(a4, w4, residual=r4)
a5, w5 = r(8192, 5120, s=0.5), r(1280, 5120, s=0.02)
cands["splitk 8192x1280x5120 +R"] = lambda: ops.gemm(a5, w5, residual=r4)
a6, w6 = r(2048, 1280, s=0.5), r(1280, 1280, s=0.03)
cands["gemm2 2048x1280x1280"] = lambda: ops.gemm(a6, w6)
xc, wc = r(F * 64 * 64, 320, s=0.5), r(320, 9 * 320, s=0.02)
cands["conv l0"] = lambda: ops.gemm(xc, wc, mode=ops.CONV_S1, geom=(64, 64, 64, 64), m_out=F * 4096)
xc2, wc2 = r(F * 16 * 16, 1280, s=0.5), r(1280, 9 * 1280, s=0.01)
cands["conv l2 splitk"] = lambda: ops.gemm(xc2, wc2, mode=ops.CONV_S1, geom=(16, 16, 16, 16), m_out=F * 256)
wg = r(2560, 320, s=0.05)
cands["gemm4 geglu"] = lambda: ops.gemm(a, wg, geglu=True)
# attention
qkv = r(F * 4096, 960, s=0.5)
cands["attn_fwd l0"] = lambda: ops.attn_fwd(qkv[:, :320], qkv[:, 320:640], qkv[:, 640:], 4096, 4096, 8, 40, F)[0]
qkv1 = r(F * 1024, 1920, s=0.5)
cands["attn_fwd l1"] = lambda: ops.attn_fwd(qkv1[:, :640], qkv1[:, 640:1280], qkv1[:, 1280:], 1024, 1024, 8, 80, F)[0]
kv = r(2 * 77, 640, s=0.5)
cands["attn_fwd cross l0"] = lambda: ops.attn_fwd(qkv[:, :320], kv[:, :320], kv[:, 320:], 4096, 77, 8, 40, F, kv_bdiv=16)[0]
qb = r(16 * 4096, 960, s=0.5)
ob, lseb = ops.attn_fwd(qb[:, :320], qb[:, 320:640], qb[:, 640:], 4096, 4096, 8, 40, 16)
dob = r(16 * 4096, 320)
def attn_bwd():
    dq, dk, dv = ops.attn_bwd(qb[:, :320], qb[:, 320:640], qb[:, 640:], ob, dob, lseb, 4096, 4096, 8, 40, 16)
    return torch.cat([dq, dk, dv], 1)
cands["attn_bwd l0"] = attn_bwd
cands["tattn_fwd l0"] = lambda: ops.tattn_fwd(qkv[:, :320], qkv[:, 320:640], qkv[:, 640:], 2, 16, 4096, 8, 40)
def tbwd():
    d = torch.empty_like(qb)
    ops.tattn_bwd(qb[:, :320], qb[:, 320:640], qb[:, 640:], dob, d[:, :320], d[:, 320:640], d[:, 640:], 1, 16, 4096, 8, 40)
    return d
cands["tattn_bwd l0"] = tbwd
# norms / glue
gam, bet = torch.ones(320, device=dev), torch.zeros(320, device=dev)
cands["gn stats+apply l0"] = lambda: ops.gn_apply(xc, None, ops.gn_stats(xc, None, F, 4096, 1e-5), gam, bet, True, F, 4096)
cands["ln fwd l0"] = lambda: ops.layernorm_fwd(xc, gam, bet)[0]
stx = ops.gn_stats(xc, None, F, 4096, 1e-5)
cands["gn bwd l0"] = lambda: ops.gn_bwd(xc, None, xc, stx, gam, bet, True, F, 4096)
ff = r(65536, 2560)
cands["geglu fwd"] = lambda: ops.geglu_fwd(ff)

noise_stream = torch.cuda.Stream()
na, nw = r(65536, 1280, s=0.5), r(1280, 1280, s=0.03)
nq = r(16 * 4096, 960, s=0.5)
def noise(n):
    with torch.cuda.stream(noise_stream):
        for i in range(n):
            ops.gemm(na, nw)
            ops.attn_fwd(nq[:, :320], nq[:, 320:640], nq[:, 640:], 4096, 4096, 8, 40, 16)
            ops.layernorm_fwd(na, torch.ones(1280, device=dev), torch.zeros(1280, device=dev))
for name, fn in cands.items():
    ref = fn().clone(); torch.cuda.synchronize()
    again = fn().clone(); torch.cuda.synchronize()
    bad = 0; worst = 0.0
    for rep in range(6):
        noise(6)
        out = fn()
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1; worst = max(worst, float((out.float() - ref.float()).abs().max()))
    print("%-34s solo rerun identical %s | under load: %d/6 differ, max diff %.4g" % (name, bool(torch.equal(again, ref)), bad, worst), flush=True)
