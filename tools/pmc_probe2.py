"""Small workload for SQ-counter passes (rocprofv3 --pmc): the tiled GEMM / conv kernels (gemm3 and gemm5 on the same
shapes) and the level-0 spatial attention, two launches each.  Prints the launch order."""
import sys
import torch
sys.path.insert(0, ".")
dev = torch.device("cuda:0")
from motionclone_amd import ops  # noqa: E402


def r(*s, sc=1.0):
    return (torch.randn(*s, device=dev) * sc).half()


what = sys.argv[1] if len(sys.argv) > 1 else "gemm"
if what == "twowg":
    # round 4: the short-K dense layers on the 8-wave 256 x 320 tiles (cfg 11) and on 256 x 160 tiles, two workgroups per CU (cfg 9)
    x1, wo1, r1 = r(32768, 640), r(640, 640, sc=0.02), r(32768, 640)
    wq1, wg1 = r(1920, 640, sc=0.02), r(5120, 640, sc=0.02)
    torch.cuda.synchronize()
    for cfg in (11, 9):
        for _ in range(2):
            ops.gemm(x1, wo1, residual=r1, cfg=cfg)          # to_out level 1 + R: one wave of 256 x 320 tiles, K = 640
            ops.gemm(x1, wq1, cfg=cfg)                       # qkv level 1
            ops.gemm(x1, wg1, geglu=True, cfg=cfg)           # FeedForward level 1, fused GEGLU
    print("order: for cfg in (11 = 8 waves, 9 = two workgroups per CU): 2 x [to_out_l1+R, qkv_l1, ff1_l1 geglu]")
elif what == "widen":
    # round 5: the wide-N short-K layers where the vendor GEMM is 21 - 31 % ahead (profiles/r05_vendor_anchor.md): gemm5 on
    # 256 x 320 tiles (cfg 11) and torch.matmul (hipBLASLt: persistent stream-K, MT 256 x 256 x 64), two launches each
    shapes = [(32768, 5120, 640), (8192, 10240, 1280), (8192, 3840, 1280)]
    ops_ = [(r(M, K), r(N, K, sc=0.02), torch.empty(M, N, device=dev, dtype=torch.float16)) for M, N, K in shapes]
    torch.cuda.synchronize()
    for x, w, o in ops_:
        for _ in range(2):
            ops.gemm(x, w, cfg=11, out=o)
        for _ in range(2):
            torch.matmul(x, w.t(), out=o)
    print("order: for (M, N, K) in %s: 2 x gemm5 <256 x 320>, 2 x torch.matmul" % (shapes,))
elif what == "widen6":
    # round 6: the same three layers - gemm5 one-pass (cfg 11), the persistent tile loop (gemm6, the library's choice), the 256 x 256
    # four-wave geometry (cfg 7) and torch.matmul (hipBLASLt), two launches each
    shapes = [(32768, 5120, 640), (8192, 10240, 1280), (8192, 3840, 1280)]
    ops_ = [(r(M, K), r(N, K, sc=0.02), torch.empty(M, N, device=dev, dtype=torch.float16)) for M, N, K in shapes]
    torch.cuda.synchronize()
    for x, w, o in ops_:
        for kw in (dict(cfg=11), dict(tileloop=True), dict(cfg=7)):
            for _ in range(2):
                ops.gemm(x, w, out=o, **kw)
        for _ in range(2):
            torch.matmul(x, w.t(), out=o)
    print("order: for (M, N, K) in %s: 2 x gemm5 <256 x 320>, 2 x gemm6 tile loop, 2 x gemm5 <256 x 256, 4 waves>, 2 x torch.matmul" % (shapes,))
elif what == "gemm":
    F = 32
    x1, wq1, wo1, r1 = r(32768, 640), r(1920, 640, sc=0.02), r(640, 640, sc=0.02), r(32768, 640)
    xc, wc = r(F * 1024, 1280), r(640, 9 * 1280, sc=0.02)
    torch.cuda.synchronize()
    for cfg in (1, 11, 12):
        for _ in range(2):
            ops.gemm(x1, wq1, cfg=cfg)                       # qkv level 1: M=32768 N=1920 K=640
            ops.gemm(x1, wo1, residual=r1, cfg=cfg)          # to_out level 1 + R: one wave of tiles, K=640
            ops.gemm(xc, wc, mode=ops.CONV_S1, geom=(32, 32, 32, 32), m_out=F * 1024, cfg=cfg)   # conv 1280->640, K=11520
    print("order: for cfg in (1 = gemm3, 11 = gemm5, 12 = gemm5 no stagger): 2 x [qkv_l1, to_out_l1+R, conv_l1 K=11520]")
else:
    Fr, N, d = 16, 4096, 40
    C = 8 * d
    qkv = r(Fr * N, 3 * C, sc=0.5)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    for _ in range(2):
        o, lse = ops.attn_fwd(q, k, v, N, N, 8, d, Fr)
    do = r(Fr * N, C)
    for _ in range(2):
        ops.attn_bwd(q, k, v, o, do, lse, N, N, 8, d, Fr)
    print("order: 2 x attn_fwd level 0 (16 f x 8 heads x 4096^2, d = 40), 2 x attn_bwd (dq, dkdv)")
torch.cuda.synchronize()
