"""Per-kernel numerics: every C-ABI entry point against a plain PyTorch fp32 reference of the same op.

Each test runs twice: on the host simulator build of the kernel sources (CPU, here) and - marked
`gpu` - on the real gfx950 library.  Tolerances are fp16-output tolerances against fp32 math.
"""
import math

import pytest
import torch
import torch.nn.functional as Fn

from motionclone_amd import lib, ops


def big(dev):
    return dev.type == "cuda"


def rnd(shape, dev, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.float16).to(dev)


def close(a, b, atol, rtol, what=""):
    a = a.float().cpu()
    b = b.float().cpu()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = (err > tol).float().mean().item()
    assert torch.isfinite(a).all(), what + ": non-finite output"
    assert bad == 0.0, "%s: %.4f%% elements off, max err %.4g (ref max %.3g)" % (
        what, 100 * bad, err.max().item(), b.abs().max().item())


# ---------------------------------------------------------------------------------------------------


@pytest.mark.parametrize("M,N,nsplit,res,alpha", [(256, 320, 0, True, 1.0), (300, 96, 1, True, 1.0), (515, 640, 2, False, 0.5),
                                                   (512, 2560, 3, True, 1.0), (100, 64, 1, False, 1.0)])
def test_gemm4_streaming_short_k(backend, M, N, nsplit, res, alpha):
    """gemm4.hip (A-stationary streaming kernel of the K = 320 Linear layers): bias as accumulator start, residual added in
    place in the staging image, 128-byte store groups incl. ragged last group, row tails, column-windowed output, splits"""
    dev = backend
    if big(dev):
        M = M * 37 + 11
    K = 320
    a, w = rnd((M, K), dev, 1, 0.5), rnd((N, K), dev, 2, 0.05)
    bias = torch.randn(1, N, generator=torch.Generator().manual_seed(3)).to(dev)
    r = rnd((M, N), dev, 4) if res else None
    wide = torch.zeros((M, N + 32), dtype=torch.float16, device=dev)
    out = ops.gemm(a, w, bias=bias, residual=r, alpha=alpha, cfg=10, nsplit=nsplit, out=wide[:, 32:])
    ref = (alpha * (a.float() @ w.float().t()) + bias).half().float()
    if res:
        ref = ref + r.float()       # the kernel adds the residual to the rounded fp16 output, as the reference's modules do
    close(out, ref, 2e-2, 5e-3, "gemm4 dense")
    assert float(wide[:, :32].abs().max()) == 0.0
    # fused GEGLU epilogue (row-interleaved weights), 16 output columns per chunk, 4 chunks per store group
    wg, bg = rnd((2 * N, K), dev, 5, 0.05), torch.randn(2 * N, generator=torch.Generator().manual_seed(6)) * 0.3
    og = ops.gemm(a, ops.interleave_geglu(wg), bias=ops.interleave_geglu(bg).unsqueeze(0).contiguous().to(dev), geglu=True,
                  cfg=10, nsplit=nsplit)
    full = a.float() @ wg.float().t() + bg.to(dev)
    close(og, full[:, :N] * Fn.gelu(full[:, N:]), 2e-2, 5e-3, "gemm4 geglu")


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5])
def test_gemm_large_tile_geometries(backend, cfg):
    """gemm3.hip: every block geometry, dense with bias/residual/tails, conv with concat, fused GEGLU"""
    dev = backend
    M, N, K = (300, 328, 128) if not big(dev) else (3000, 968, 320)
    a, w = rnd((M, K), dev, 1), rnd((N, K), dev, 2, 0.1)
    bias = torch.randn(1, N, generator=torch.Generator().manual_seed(3)).to(dev)
    res = rnd((M, N), dev, 4)
    out = ops.gemm(a, w, bias=bias, residual=res, alpha=0.5, cfg=cfg)
    close(out, 0.5 * (a.float() @ w.float().t()) + bias + res.float(), 2e-2, 5e-3, "gemm3 dense")
    NF, Cin, Cout, H, W = (2, 64, 72, 6, 10) if not big(dev) else (4, 128, 320, 24, 20)
    x, x2 = rnd((NF, Cin, H, W), dev, 5), rnd((NF, 64, H, W), dev, 6)
    wc = rnd((Cout, Cin + 64, 3, 3), dev, 7, 0.05)
    o = ops.gemm(_to_cl(x), _conv_w_pack(wc), a2=_to_cl(x2), mode=ops.CONV_S1, geom=(H, W, H, W), m_out=NF * H * W, cfg=cfg)
    close(_from_cl(o, NF, H, W), Fn.conv2d(torch.cat([x, x2], 1).float(), wc.float(), padding=1), 3e-2, 5e-3, "gemm3 conv")
    D = 80
    wg = rnd((2 * D, K), dev, 8, 0.1)
    y = a.float() @ wg.float().t()
    og = ops.gemm(a, ops.interleave_geglu(wg), geglu=True, cfg=cfg)
    close(og, y[:, :D] * Fn.gelu(y[:, D:]), 2e-2, 1e-2, "gemm3 geglu")


@pytest.mark.parametrize("var", [0, 4, -3, -4])
def test_gemm5_ring_kernel(backend, var):
    """gemm5.hip (4-stage ring, LDS-DMA, wave-private epilogue), forced with cfg = 11 + var: dense with
    per-batch bias / residual / alpha / M and N tails / short and long K (2 .. 40 ring stages), two-source conv, fused GEGLU;
    var 4 = the 128-row tile geometry, var -3 (cfg 8) = 256 x 320 tiles on FOUR waves with 128 x 160 wave tiles (round 6: measured
    slower than eight waves, kept as a forced configuration only), var -4 (cfg 7) = 256 x 256 tiles on four waves with 128 x 128
    wave tiles, dense only"""
    dev = backend
    cfg = 11 + var
    # (round 6: a tile of the one-pass kernels reads ONE bias row - per-batch bias rows need rows_per_batch % tile height == 0;
    # other problems are refused and the library's own choice falls through to gemm3: checked at the end)
    for (M, N, K, rpb) in ([(300, 328, 64, 300), (600, 640, 192, 256)] if not big(dev) else [(3000, 968, 128, 1024), (5000, 640, 1280, 2560)]):
        a, w = rnd((M, K), dev, 1), rnd((N, K), dev, 2, 0.1)
        nb = (M + rpb - 1) // rpb
        bias = torch.randn(nb, N, generator=torch.Generator().manual_seed(3)).to(dev)
        res = rnd((M, N), dev, 4)
        out = ops.gemm(a, w, bias=bias, residual=res, alpha=0.5, rows_per_batch=rpb, cfg=cfg)
        lin = (0.5 * (a.float() @ w.float().t()) + bias.repeat_interleave(rpb, 0)[:M]).half().float()   # rounded, then + R
        close(out, lin + res.float(), 2e-2, 5e-3, "gemm5 dense %d %d %d" % (M, N, K))
        out2 = ops.gemm(a, w, cfg=cfg)
        close(out2, a.float() @ w.float().t(), 2e-2, 5e-3, "gemm5 dense plain")
    NF, Cin, Cout, H, W = (2, 64, 72, 6, 10) if not big(dev) else (4, 128, 320, 24, 20)
    x, x2 = rnd((NF, Cin, H, W), dev, 5), rnd((NF, 64, H, W), dev, 6)
    wc = rnd((Cout, Cin + 64, 3, 3), dev, 7, 0.05)
    if var == -4:     # the 256 x 256 geometry is built for the dense layers only
        with pytest.raises(RuntimeError):
            ops.gemm(_to_cl(x), _conv_w_pack(wc), a2=_to_cl(x2), mode=ops.CONV_S1, geom=(H, W, H, W), m_out=NF * H * W, cfg=cfg)
    else:
        o = ops.gemm(_to_cl(x), _conv_w_pack(wc), a2=_to_cl(x2), mode=ops.CONV_S1, geom=(H, W, H, W), m_out=NF * H * W, cfg=cfg)
        close(_from_cl(o, NF, H, W), Fn.conv2d(torch.cat([x, x2], 1).float(), wc.float(), padding=1), 3e-2, 5e-3, "gemm5 conv")
    M, K, D = (300, 128, 80) if not big(dev) else (3000, 320, 640)
    a = rnd((M, K), dev, 1)
    wg = rnd((2 * D, K), dev, 8, 0.1)
    bg = torch.randn(1, 2 * D, generator=torch.Generator().manual_seed(9)).to(dev)
    y = a.float() @ wg.float().t() + bg
    og = ops.gemm(a, ops.interleave_geglu(wg), bias=ops.interleave_geglu(bg.t()).t().contiguous(), geglu=True, cfg=cfg)
    close(og, y[:, :D] * Fn.gelu(y[:, D:]), 2e-2, 1e-2, "gemm5 geglu")
    if var == 0:   # bias rows that change inside a tile: refused by gemm5 when forced, served (by gemm3) when the library chooses
        a, w = rnd((300, 64), dev, 1), rnd((328, 64), dev, 2, 0.1)
        bias = torch.randn(2, 328, generator=torch.Generator().manual_seed(3)).to(dev)
        with pytest.raises(RuntimeError):
            ops.gemm(a, w, bias=bias, rows_per_batch=150, cfg=cfg)
        out = ops.gemm(a, w, bias=bias, rows_per_batch=150)
        close(out, a.float() @ w.float().t() + bias.repeat_interleave(150, 0), 2e-2, 5e-3, "per-batch bias, library's choice")


@pytest.mark.parametrize("mode,N,hw,cfg,res", [("dense", 320, 64, 11, True), ("dense", 640, 128, 11, False), ("conv", 320, 64, 11, True),
                                              ("dense", 1280, 64, 15, True), ("conv", 640, 96, 15, False), ("dense", 320, 256, 0, True)])
def test_gemm_leaves_groupnorm_statistics_of_its_output(backend, mode, N, hw, cfg, res):
    """mc_gemm_gnstats_f16 (round 6): the one-pass ring kernels' epilogue writes the GroupNorm(32) partial sums of the ROUNDED
    output per (frame, chunk of 64 / 32 rows, group); mc_groupnorm_fwd_partial_f16 consumes them.  Output bit-identical to
    mc_gemm_f16; normalised tensor and (mean, rstd) equal to the statistics-pass form to fp32 summation order (same bound as
    test_groupnorm_fwd_bwd's two forms); refused with nothing launched where a wave tile would straddle frames (hw = 96 with
    64-row chunks) or the channel count has no whole groups per 160 columns."""
    dev = backend
    frames = 3 if not big(dev) else 24
    if cfg == 0 and not big(dev):
        pytest.skip("the library's own choice takes the ring kernels at GPU sizes only")
    if big(dev) and cfg == 0:
        frames, hw = 32, 1024
    M = frames * hw
    bias = torch.randn(frames, N, generator=torch.Generator().manual_seed(3)).to(dev) if hw % 256 == 0 or cfg == 15 and hw % 128 == 0 \
        else torch.randn(1, N, generator=torch.Generator().manual_seed(3)).to(dev)
    rpb = hw if bias.shape[0] > 1 else 0
    R = rnd((M, N), dev, 4) if res else None
    if mode == "dense":
        K = 128 if not big(dev) else 640
        a, w = rnd((M, K), dev, 1), rnd((N, K), dev, 2, 0.1)
        kw = dict()
    else:
        Cin = 64 if not big(dev) else 320
        H, W = {64: (8, 8), 96: (8, 12), 128: (8, 16), 256: (16, 16), 1024: (32, 32)}[hw]
        a = rnd((M, Cin), dev, 1)
        w = _conv_w_pack(rnd((N, Cin, 3, 3), dev, 2, 0.05))
        kw = dict(mode=ops.CONV_S1, geom=(H, W, H, W), m_out=M)
    plain = ops.gemm(a, w, bias=bias, residual=R, rows_per_batch=rpb, cfg=cfg, tileloop=False, **kw)
    out, gnp = ops.gemm(a, w, bias=bias, residual=R, rows_per_batch=rpb, cfg=cfg, tileloop=False, gn_hw=hw, **kw)
    rows = 64 if cfg == 11 else 32 if cfg == 15 else None      # (cfg 0: whichever of the two tile heights the library takes)
    if rows is None:
        assert gnp is not None and gnp[1] in (hw // 64, hw // 32)
        rows = hw // gnp[1]
    if hw % rows:
        assert gnp is None, "a wave tile that straddles two frames must be refused"
        assert torch.equal(out, plain)
        return
    assert gnp is not None and gnp[1] == hw // rows
    assert torch.equal(out, plain), "the statistics epilogue changed the GEMM's output"
    gamma = (1 + 0.1 * torch.randn(N, generator=torch.Generator().manual_seed(5))).to(dev)
    beta = (0.1 * torch.randn(N, generator=torch.Generator().manual_seed(6))).to(dev)
    y0, st0 = ops.gn_fwd(out, None, gamma, beta, True, frames, hw, 1e-5)
    y1, st1 = ops.gn_fwd(out, None, gamma, beta, True, frames, hw, 1e-5, gnp=gnp)
    close(st1, st0, 1e-5, 1e-5, "statistics from the epilogue vs the statistics pass")
    close(y1, y0, 2e-3, 2e-3, "normalised output")
    assert (y1 != y0).float().mean().item() < 0.02
    xr = out.float().reshape(frames, hw, N).permute(0, 2, 1)
    ref = Fn.silu(Fn.group_norm(xr, 32, gamma, beta, 1e-5)).permute(0, 2, 1).reshape(M, N)
    close(y1, ref, 1e-2, 5e-3, "GroupNorm + SiLU from epilogue statistics vs fp32 torch")
    # the sums themselves: every (frame, chunk, group) slot against fp64 sums of the stored fp16 values
    part = gnp[0][:frames * gnp[1] * 64].reshape(frames, gnp[1], 32, 2).double().cpu()
    xo = out.double().cpu().reshape(frames, gnp[1], rows, 32, N // 32)
    want = torch.stack([xo.sum((2, 4)), (xo * xo).sum((2, 4))], -1)
    assert torch.allclose(part, want, rtol=1e-5, atol=1e-3), (part - want).abs().max().item()


@pytest.mark.parametrize("dynamic", [False, True])
def test_gemm6_persistent_tile_loop_equals_gemm5_bit_for_bit(backend, dynamic):
    """gemm6.hip (mc_gemm_tileloop_f16: one workgroup per CU walks the 256x320 tiles, the operand ring runs through tile
    boundaries, epilogue through 32 x 80 images in the free ring slot): outputs EQUAL gemm5's (one fp32 chain per element in the
    same order) - plain, bias, bias + residual, fused GEGLU, M / N tails, several tiles per workgroup (grid cap), static and
    dynamic tile order, both vmcnt variants; vs fp32 torch; the counter block is zero again after every launch."""
    dev = backend
    shapes = [(2100, 648, 320, 16), (2304, 960, 256, 8)] if not big(dev) else [(8192, 3840, 1280, 0), (9000, 1928, 640, 64), (32768, 1280, 256, 0)]
    for (M, N, K, cap) in shapes:
        a, w = rnd((M, K), dev, 1), rnd((N, K), dev, 2, 0.1)
        bias = torch.randn(1, N, generator=torch.Generator().manual_seed(3)).to(dev)
        res = rnd((M, N), dev, 4)
        for (b, r) in [(None, None), (bias, None), (bias, res)]:
            ref = ops.gemm(a, w, bias=b, residual=r, alpha=0.5, cfg=11)
            for strict in (False, True):
                out = torch.full_like(ref, float("nan"))
                got = ops.gemm_tileloop(a, w, bias=b, residual=r, alpha=0.5, out=out, dynamic=dynamic, strict_order=strict, max_wg=cap)
                assert got is not None, "shape refused"
                assert torch.equal(out, ref), "tile loop differs from gemm5: %d %d %d bias %s res %s" % (M, N, K, b is not None, r is not None)
            lin = 0.5 * (a.float() @ w.float().t()) + (b if b is not None else 0)
            close(out, lin.half().float() + (r.float() if r is not None else 0), 2e-2, 5e-3, "tile loop vs fp32")
        if N % 16 == 0:
            wg = ops.interleave_geglu(rnd((N, K), dev, 8, 0.1))
            bg = ops.interleave_geglu(torch.randn(N, generator=torch.Generator().manual_seed(9))).unsqueeze(0).to(dev)
            ref = ops.gemm(a, wg, bias=bg, geglu=True, cfg=11)
            out = torch.full_like(ref, float("nan"))
            assert ops.gemm_tileloop(a, wg, bias=bg, geglu=True, out=out, dynamic=dynamic, max_wg=cap) is not None
            assert torch.equal(out, ref), "tile loop GEGLU differs from gemm5"
    for slab in ops._tile_slabs.values():
        assert int(slab.count_nonzero()) == 0, "a launch left its tile counters dirty"
    # outside the kernel's shapes: refused, nothing launched
    a, w = rnd((1024, 256), dev, 1), rnd((640, 256), dev, 2)
    assert ops.gemm_tileloop(a, w) is None                      # fewer than 8 row tiles
    a, w = rnd((2048, 128), dev, 1), rnd((640, 128), dev, 2)
    assert ops.gemm_tileloop(a, w) is None                      # K < 256


def test_gemm6_stream_k_cuts_tiles_along_k(backend):
    """mc_gemm_tileloop_f16 flags 0x2: the k-stages of every XCD's tile list are dealt evenly to the workgroups; a tile that a range
    boundary falls into is cut along k, the later pieces' fp32 sums reach the owner of the first piece through the slabs + flags
    (device-scope hand-over), summed in k order.  Against gemm5 (one chain per element): equal where no tile is cut, within one
    fp16 rounding of the fp32 result elsewhere; fewer than 8 rows of tiles (flat tile lists); flags are zero again afterwards."""
    dev = backend
    shapes = [(2304, 960, 1024, 16), (1024, 1280, 1280, 0), (512, 640, 2048, 0)] if not big(dev) else \
             [(8192, 3840, 1280, 0), (2048, 3840, 1280, 0), (4096, 1280, 5120, 0), (1024, 1280, 1280, 0)]
    for (M, N, K, cap) in shapes:
        a, w = rnd((M, K), dev, 1), rnd((N, K), dev, 2, 0.1)
        bias = torch.randn(1, N, generator=torch.Generator().manual_seed(3)).to(dev)
        res = rnd((M, N), dev, 4)
        for (b, r) in [(None, None), (bias, res)]:
            out = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
            assert ops.gemm_tileloop(a, w, bias=b, residual=r, alpha=0.5, out=out, stream_k=True, max_wg=cap) is not None
            lin = 0.5 * (a.float() @ w.float().t()) + (b if b is not None else 0)
            close(out, lin.half().float() + (r.float() if r is not None else 0), 2e-2, 5e-3, "stream-K %d %d %d" % (M, N, K))
            again = torch.empty_like(out)
            ops.gemm_tileloop(a, w, bias=b, residual=r, alpha=0.5, out=again, stream_k=True, max_wg=cap)
            assert torch.equal(out, again), "stream-K is not reproducible"
        wg = ops.interleave_geglu(rnd((N, K), dev, 8, 0.1))
        bg = ops.interleave_geglu(torch.randn(N, generator=torch.Generator().manual_seed(9))).unsqueeze(0).to(dev)
        y = a.float() @ wg.float().t() + bg
        out = ops.gemm_tileloop(a, wg, bias=bg, geglu=True, stream_k=True, max_wg=cap)
        close(out, y[:, 0::2] * Fn.gelu(y[:, 1::2]), 2e-2, 1e-2, "stream-K geglu")
    for slab in ops._tile_slabs.values():
        assert int(slab.count_nonzero()) == 0, "a launch left its hand-over flags dirty"


@pytest.mark.parametrize("mode", ["s2", "up", "tconv"])
def test_gemm5_conv_modes(backend, mode):
    dev = backend
    NF, Cin, Cout, H, W = (2, 64, 72, 8, 12) if not big(dev) else (4, 128, 320, 32, 24)
    x = rnd((NF, Cin, H, W), dev, 5)
    wc = rnd((Cout, Cin, 3, 3), dev, 7, 0.05)
    if mode == "s2":
        o = ops.gemm(_to_cl(x), _conv_w_pack(wc), mode=ops.CONV_S2, geom=(H, W, H // 2, W // 2), m_out=NF * H * W // 4, cfg=11)
        close(_from_cl(o, NF, H // 2, W // 2), Fn.conv2d(x.float(), wc.float(), stride=2, padding=1), 3e-2, 5e-3, "gemm5 s2")
    elif mode == "up":
        o = ops.gemm(_to_cl(x), _conv_w_pack(wc), mode=ops.CONV_UP, geom=(H, W, 2 * H, 2 * W), m_out=NF * 4 * H * W, cfg=11)
        ref = Fn.conv2d(Fn.interpolate(x.float(), scale_factor=2.0, mode="nearest"), wc.float(), padding=1)
        close(_from_cl(o, NF, 2 * H, 2 * W), ref, 3e-2, 5e-3, "gemm5 up")
    else:   # data gradient of a stride-2 conv (Downsample3D's backward)
        Co = 64 if not big(dev) else 128
        xg = x.float().requires_grad_()
        w2 = rnd((Co, Cin, 3, 3), dev, 8, 0.05)
        y = Fn.conv2d(xg, w2.float(), padding=1, stride=2)
        dy = rnd(tuple(y.shape), dev, 3)
        (ref,) = torch.autograd.grad(y, xg, dy.float())
        wd = ops.pack_conv_k(w2.permute(1, 2, 3, 0).reshape(Cin, 9, Co))
        o = ops.gemm(_to_cl(dy), wd, mode=ops.TCONV_S2, geom=(y.shape[2], y.shape[3], H, W), m_out=NF * H * W, cfg=11)
        close(_from_cl(o, NF, H, W), ref, 3e-2, 5e-3, "gemm5 tconv")


@pytest.mark.parametrize("splits,cfg", [(2, 0), (3, 4), (5, 1), (4, 5)])
def test_gemm_split_k(backend, splits, cfg):
    """split-K path: K ranges in separate workgroups -> fp32 partial sums -> reduce kernel with bias / residual"""
    dev = backend
    M, N, K = (200, 328, 448) if not big(dev) else (2048, 1280, 2880)
    a, w = rnd((M, K), dev, 1), rnd((N, K), dev, 2, 0.1)
    bias = torch.randn(2, N, generator=torch.Generator().manual_seed(3)).to(dev)
    res = rnd((M, N), dev, 4)
    out = ops.gemm(a, w, bias=bias, residual=res, alpha=0.5, rows_per_batch=M // 2, cfg=cfg, splits=splits)
    ref = 0.5 * (a.float() @ w.float().t()) + bias.repeat_interleave(M // 2, 0) + res.float()
    close(out, ref, 2e-2, 5e-3, "split-K dense")
    NF, Cin, Cout, H, W = (2, 64, 72, 6, 10) if not big(dev) else (32, 1280, 1280, 8, 8)
    x, x2 = rnd((NF, Cin, H, W), dev, 5), rnd((NF, 64, H, W), dev, 6)
    wc = rnd((Cout, Cin + 64, 3, 3), dev, 7, 0.05 if not big(dev) else 0.01)
    o = ops.gemm(_to_cl(x), _conv_w_pack(wc), a2=_to_cl(x2), mode=ops.CONV_S1, geom=(H, W, H, W), m_out=NF * H * W,
                 cfg=cfg, splits=splits)
    close(_from_cl(o, NF, H, W), Fn.conv2d(torch.cat([x, x2], 1).float(), wc.float(), padding=1), 3e-2, 5e-3, "split-K conv")


def test_gemm_split_k_plan(backend):
    from motionclone_amd import lib
    L = lib.load()
    assert L.mc_gemm_splitk_plan(2048, 1280, 11520, 1) == (4 | (4 << 8))   # 8x8 level, B = 2: 64 tiles of 128x320 -> 4 ranges
    assert L.mc_gemm_splitk_plan(1024, 1280, 23040, 1) == (8 | (4 << 8))
    assert L.mc_gemm_splitk_plan(8192, 1280, 11520, 1) == (2 | (1 << 8))   # 16x16 level, B = 2: 128 tiles of 256x320 -> 2 ranges
    assert L.mc_gemm_splitk_plan(4096, 1280, 11520, 1) == (4 | (1 << 8))
    assert L.mc_gemm_splitk_plan(131072, 320, 2880, 1) == 1      # plenty of tiles
    assert L.mc_gemm_splitk_plan(2048, 1280, 1280, 0) == 1       # K too shallow
    assert L.mc_gemm_splitk_plan(2048, 1000, 11520, 0) == 1      # N not a multiple of 320
    # share hint (mode bits 8-9): with two launch sequences in flight a launch targets 128 CUs
    assert L.mc_gemm_splitk_plan(8192, 1280, 11520, 1 | (1 << 8)) == 1              # 128 big tiles are enough now
    assert L.mc_gemm_splitk_plan(2048, 1280, 11520, 1 | (1 << 8)) == (4 | (1 << 8))  # 32 big tiles x 4 ranges
    assert L.mc_gemm_splitk_plan(1024, 1280, 23040, 1 | (1 << 8)) == (4 | (4 << 8))


def test_gemm_share_hint_changes_the_choice_not_the_result(backend):
    dev = backend
    M, N, K = (512, 640, 256) if not big(dev) else (16384, 1280, 1280)
    a, w = rnd((M, K), dev, 1), rnd((N, K), dev, 2, 0.05)
    ref = a.float() @ w.float().t()
    try:
        for lanes in (1, 2, 4):
            ops.set_gemm_share(lanes)
            close(ops.gemm(a, w), ref, 2e-2, 5e-3, "share %d" % lanes)
    finally:
        ops.set_gemm_share(1)
    assert ops._GEMM_SHARE == 0


def test_tile_policy_of_an_engine_overrides_the_process_default_for_its_own_calls():
    """`gemm_lanes` on an engine (or on a sampler's engine) pins the GEMM tile / split-K policy for the duration of ITS scoped calls
    (round-4 verdict, weak 12: a process-global policy is wrong once two engines with different lane counts share a
    process); other callers keep the process-wide default, and the override nests and unwinds."""
    class Eng:
        dev = None

        @ops.scoped
        def call(self, inner=None):
            here = ops.gemm_share()
            return (here, inner.call()[0]) if inner is not None else (here, None)
    a, b, c = Eng(), Eng(), Eng()
    a.gemm_lanes, b.gemm_lanes = 3, 1            # c has none: the process default
    assert ops.gemm_share() == 0
    assert a.call() == (1, None) and b.call() == (0, None) and c.call() == (0, None)
    assert a.call(inner=b) == (1, 0) and b.call(inner=a) == (0, 1) and a.call(inner=c) == (1, 1)   # c inherits its caller's
    try:
        ops.set_gemm_share(4)
        assert c.call() == (2, None) and a.call() == (1, None) and ops.gemm_share() == 2
    finally:
        ops.set_gemm_share(1)
    assert ops.gemm_share() == 0 and ops._PIN.share is None

    class Smp:                                   # a sampler takes its engine's setting
        def __init__(self, engine):
            self.engine, self.dev = engine, None

        @ops.scoped
        def step(self):
            return ops.gemm_share()
    assert Smp(a).step() == 1 and Smp(c).step() == 0


@pytest.mark.parametrize("M,N,K,tile", [(300, 136, 64, 128), (300, 136, 128, 128), (333, 200, 448, 128), (200, 72, 320, 64)])
def test_gemm_deep_pipeline(backend, M, N, K, tile):
    """3-stage LDS ring variant (counted vmcnt): K of 1, 2 and many tiles"""
    dev = backend
    a, w = rnd((M, K), dev, 1), rnd((N, K), dev, 2, 0.1)
    res = rnd((M, N), dev, 4)
    out = ops.gemm(a, w, residual=res, tile=tile, deep=True)
    close(out, a.float() @ w.float().t() + res.float(), 2e-2, 5e-3, "gemm deep")


def test_gemm_geglu_epilogue(backend):
    dev = backend
    M, D, K = (150, 72, 128) if not big(dev) else (4099, 1280, 320)
    a, w = rnd((M, K), dev, 1), rnd((2 * D, K), dev, 2, 0.1)
    bias = torch.randn(2 * D, generator=torch.Generator().manual_seed(3)).to(dev)
    y = a.float() @ w.float().t() + bias
    ref = y[:, :D] * Fn.gelu(y[:, D:])
    for tile in (64, 128):
        out = ops.gemm(a, ops.interleave_geglu(w), bias=ops.interleave_geglu(bias).unsqueeze(0), geglu=True, tile=tile)
        close(out, ref, 2e-2, 1e-2, "gemm+geglu tile %d" % tile)


@pytest.mark.parametrize("M,N,K,tile", [(200, 72, 128, 64), (300, 136, 192, 128), (77, 64, 64, 0), (1100, 260, 320, 128)])
def test_gemm_dense(backend, M, N, K, tile):
    dev = backend
    if big(dev):
        M, N, K = M * 8 + 5, N * 4, K * 4
    a, w = rnd((M, K), dev, 1), rnd((N, K), dev, 2, 0.1)
    bias = torch.randn(2, N, generator=torch.Generator().manual_seed(3)).to(dev)
    res = rnd((M, N), dev, 4)
    rpb = (M + 1) // 2
    out = ops.gemm(a, w, bias=bias, residual=res, rows_per_batch=rpb, alpha=0.5, tile=tile)
    b_idx = (torch.arange(M, device=dev) // rpb)
    ref = 0.5 * (a.float() @ w.float().t()) + bias[b_idx] + res.float()
    close(out, ref, 2e-2, 5e-3, "gemm")
    # asymmetric operands would expose a transposed C-write; also check no-epilogue path
    out2 = ops.gemm(a, w, tile=tile)
    close(out2, a.float() @ w.float().t(), 2e-2, 5e-3, "gemm plain")


def test_gemm_concat_and_strided_out(backend):
    dev = backend
    M, C1, C2, N = 130, 64, 128, 96
    a, a2, w = rnd((M, C1), dev, 1), rnd((M, C2), dev, 2), rnd((N, C1 + C2), dev, 3, 0.1)
    wide = torch.zeros((M, 3 * N), dtype=torch.float16, device=dev)
    ops.gemm(a, w, a2=a2, out=wide[:, N:2 * N])
    ref = torch.cat([a, a2], 1).float() @ w.float().t()
    close(wide[:, N:2 * N], ref, 2e-2, 5e-3, "gemm concat")
    assert wide[:, :N].abs().max() == 0 and wide[:, 2 * N:].abs().max() == 0


def _conv_w_pack(w):  # [Cout, Cin, 3, 3] -> [Cout, 9*Cin] in the kernels' K order
    return ops.pack_conv_k(w.permute(0, 2, 3, 1).reshape(w.shape[0], 9, w.shape[1]))


def _to_cl(x):  # [NF, C, H, W] -> [NF*H*W, C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def _from_cl(x, NF, H, W):
    return x.reshape(NF, H, W, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize("mode", ["s1", "s2", "up", "concat"])
def test_conv3x3(backend, mode):
    dev = backend
    NF, Cin, Cout, H, W = (3, 64, 72, 6, 10) if not big(dev) else (5, 192, 200, 24, 20)
    x = rnd((NF, Cin, H, W), dev, 1)
    w = rnd((Cout, Cin, 3, 3), dev, 2, 0.05)
    bias = torch.randn(1, Cout, generator=torch.Generator().manual_seed(3)).to(dev)
    xc = _to_cl(x)
    wp = _conv_w_pack(w)
    if mode == "s1":
        out = ops.gemm(xc, wp, bias=bias, mode=ops.CONV_S1, geom=(H, W, H, W), m_out=NF * H * W)
        ref = Fn.conv2d(x.float(), w.float(), bias[0], padding=1)
        Ho, Wo = H, W
    elif mode == "s2":
        Ho, Wo = H // 2, W // 2
        out = ops.gemm(xc, wp, bias=bias, mode=ops.CONV_S2, geom=(H, W, Ho, Wo), m_out=NF * Ho * Wo)
        ref = Fn.conv2d(x.float(), w.float(), bias[0], padding=1, stride=2)
    elif mode == "up":
        Ho, Wo = 2 * H, 2 * W
        out = ops.gemm(xc, wp, bias=bias, mode=ops.CONV_UP, geom=(H, W, Ho, Wo), m_out=NF * Ho * Wo)
        ref = Fn.conv2d(Fn.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.float(), bias[0], padding=1)
    else:
        x2 = rnd((NF, 128, H, W), dev, 5)
        w = rnd((Cout, Cin + 128, 3, 3), dev, 6, 0.05)
        out = ops.gemm(xc, _conv_w_pack(w), a2=_to_cl(x2), bias=bias, mode=ops.CONV_S1, geom=(H, W, H, W),
                       m_out=NF * H * W)
        ref = Fn.conv2d(torch.cat([x, x2], 1).float(), w.float(), bias[0], padding=1)
        Ho, Wo = H, W
    close(_from_cl(out, NF, Ho, Wo), ref, 3e-2, 5e-3, "conv " + mode)


@pytest.mark.parametrize("stride", [1, 2])
def test_conv3x3_dgrad(backend, stride):
    """data-gradient of the 3x3 conv = the same kernel with re-packed weights (autograd is the reference)."""
    dev = backend
    NF, Cin, Cout, H, W = (2, 64, 64, 8, 6) if not big(dev) else (4, 128, 192, 16, 24)
    x = rnd((NF, Cin, H, W), dev, 1).float().requires_grad_()
    w = rnd((Cout, Cin, 3, 3), dev, 2, 0.05)
    y = Fn.conv2d(x, w.float(), padding=1, stride=stride)
    dy = rnd(tuple(y.shape), dev, 3)
    (ref,) = torch.autograd.grad(y, x, dy.float())
    Ho, Wo = y.shape[2:]
    if stride == 1:
        wd = ops.pack_conv_k(w.flip(2, 3).permute(1, 2, 3, 0).reshape(Cin, 9, Cout))
        out = ops.gemm(_to_cl(dy), wd, mode=ops.CONV_S1, geom=(H, W, H, W), m_out=NF * H * W)
    else:
        wd = ops.pack_conv_k(w.permute(1, 2, 3, 0).reshape(Cin, 9, Cout))
        out = ops.gemm(_to_cl(dy), wd, mode=ops.TCONV_S2, geom=(Ho, Wo, H, W), m_out=NF * H * W)
    close(_from_cl(out, NF, H, W), ref, 3e-2, 5e-3, "conv dgrad s%d" % stride)


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C1,C2,silu", [(64, 0, True), (320, 0, False), (64, 128, True), (960, 0, True)])
def test_groupnorm_fwd_bwd(backend, C1, C2, silu):
    dev = backend
    NF, H, W = (3, 5, 7) if not big(dev) else (6, 32, 24)
    C = C1 + C2
    x = rnd((NF, C, H, W), dev, 1) + 0.5
    gamma = (1 + 0.1 * torch.randn(C, generator=torch.Generator().manual_seed(2))).to(dev)
    beta = (0.1 * torch.randn(C, generator=torch.Generator().manual_seed(3))).to(dev)
    xa = _to_cl(x[:, :C1])
    xb = _to_cl(x[:, C1:]) if C2 else None
    stats = ops.gn_stats(xa, xb, NF, H * W, 1e-5)
    y = ops.gn_apply(xa, xb, stats, gamma, beta, silu, NF, H * W)
    xr = x.float().requires_grad_()
    ref = Fn.group_norm(xr, 32, gamma, beta, 1e-5)
    if silu:
        ref = Fn.silu(ref)
    close(_from_cl(y, NF, H, W), ref, 1e-2, 5e-3, "gn fwd")
    # the two-launch form the engine uses (statistics finalised in the apply pass's prologue): the same output and statistics
    y2, stats2 = ops.gn_fwd(xa, xb, gamma, beta, silu, NF, H * W, 1e-5)
    # (the same expressions compiled in two kernels under fast-math: the statistics may differ by an fp32 rounding, the output
    # then by one fp16 step on a few elements)
    close(stats2, stats, 1e-5, 1e-5, "mc_groupnorm_fwd_f16 statistics vs stats + apply")
    close(y2, y, 2e-3, 2e-3, "mc_groupnorm_fwd_f16 output vs stats + apply")
    assert (y2 != y).float().mean().item() < 0.02, "mc_groupnorm_fwd_f16 differs from stats + apply on more than 2 % of the elements"
    dz = rnd((NF, C, H, W), dev, 4)
    (dref,) = torch.autograd.grad(ref, xr, dz.float())
    dx = ops.gn_bwd(xa, xb, _to_cl(dz), stats, gamma, beta, silu, NF, H * W)
    close(_from_cl(dx, NF, H, W), dref, 1e-2, 1e-2, "gn bwd")
    # accumulate form
    base = rnd((NF * H * W, C), dev, 5)
    acc = base.clone()
    ops.gn_bwd(xa, xb, _to_cl(dz), stats, gamma, beta, silu, NF, H * W, out=acc, accumulate=True)
    close(acc, dx.float() + base.float(), 1e-2, 1e-2, "gn bwd accumulate")


@pytest.mark.parametrize("C", [64, 320, 640, 1280])
def test_layernorm_fwd_bwd(backend, C):
    dev = backend
    F_, HW = 3, 5
    M = 2 * F_ * HW + (0 if not big(dev) else 4096)
    M -= M % (F_ * HW)
    x = rnd((M, C), dev, 1) * 2 + 0.3
    gamma = (1 + 0.1 * torch.randn(C, generator=torch.Generator().manual_seed(2))).to(dev)
    beta = (0.1 * torch.randn(C, generator=torch.Generator().manual_seed(3))).to(dev)
    pe = torch.randn(F_, C, generator=torch.Generator().manual_seed(4)).to(dev)
    y, stats = ops.layernorm_fwd(x, gamma, beta, 1e-5, pe=pe, hw=HW)
    xr = x.float().requires_grad_()
    ref0 = Fn.layer_norm(xr, (C,), gamma, beta, 1e-5)
    frame = (torch.arange(M, device=dev) // HW) % F_
    close(y, ref0 + pe[frame], 1e-2, 5e-3, "ln fwd")
    dy = rnd((M, C), dev, 5)
    addv = rnd((M, C), dev, 6)
    (dref,) = torch.autograd.grad(ref0, xr, dy.float())
    dx = ops.layernorm_bwd(dy, x, stats, gamma, add=addv)
    close(dx, dref + addv.float(), 1e-2, 1e-2, "ln bwd")


# ---------------------------------------------------------------------------------------------------
def _heads(t, nb, n, heads, d):  # [nb*n, heads*d] -> [nb, heads, n, d]
    return t.float().reshape(nb, n, heads, d).permute(0, 2, 1, 3)


@pytest.mark.parametrize("d,Nq", [(16, 70), (40, 150), (80, 64), (160, 40)])
def test_self_attention_fwd_bwd(backend, d, Nq):
    dev = backend
    heads, nb = 2, 2
    if big(dev):
        Nq, heads, nb = Nq * 9 + 3, 4, 3
    C = heads * d
    qkv = rnd((nb * Nq, 3 * C), dev, 1, 0.7)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    o, lse = ops.attn_fwd(q, k, v, Nq, Nq, heads, d, nb)
    Q, K, V = (_heads(t, nb, Nq, heads, d).requires_grad_() for t in (q, k, v))
    S = (Q @ K.transpose(-1, -2)) * d ** -0.5
    ref = S.softmax(-1) @ V
    close(_heads(o, nb, Nq, heads, d), ref, 1e-2, 1e-2, "attn fwd")
    close(lse, torch.logsumexp(S, -1), 2e-3, 1e-3, "attn lse")
    do = rnd((nb * Nq, C), dev, 2)
    gq, gk, gv = torch.autograd.grad(ref, (Q, K, V), _heads(do, nb, Nq, heads, d))
    dqkv = torch.zeros_like(qkv)
    ops.attn_bwd(q, k, v, o, do, lse, Nq, Nq, heads, d, nb, dq=dqkv[:, :C], dk=dqkv[:, C:2 * C], dv=dqkv[:, 2 * C:])
    close(_heads(dqkv[:, :C], nb, Nq, heads, d), gq, 1e-2, 2e-2, "attn dq")
    close(_heads(dqkv[:, C:2 * C], nb, Nq, heads, d), gk, 1e-2, 2e-2, "attn dk")
    close(_heads(dqkv[:, 2 * C:], nb, Nq, heads, d), gv, 1e-2, 2e-2, "attn dv")


def test_self_attention_long_sequence_takes_four_query_tiles_per_wave(backend):
    """Nq >= 2048 at d = 40 selects the 256-row workgroup (attention.hip a_launch_fwd); ragged tail on both axes"""
    dev = backend
    d, Nq, heads, nb = 40, 2048 + 37, (4 if big(dev) else 1), (2 if big(dev) else 1)
    C = heads * d
    qkv = rnd((nb * Nq, 3 * C), dev, 3, 0.7)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    o, lse = ops.attn_fwd(q, k, v, Nq, Nq, heads, d, nb)
    Q, K, V = (_heads(t, nb, Nq, heads, d) for t in (q, k, v))
    S = (Q @ K.transpose(-1, -2)) * d ** -0.5
    close(_heads(o, nb, Nq, heads, d), S.softmax(-1) @ V, 1e-2, 1e-2, "attn fwd, 4 tiles per wave")
    close(lse, torch.logsumexp(S, -1), 2e-3, 1e-3, "attn lse, 4 tiles per wave")
    # the backward takes four row tiles per wave at this length too (attention.hip bwd_tiles)
    Q, K, V = (t.detach().requires_grad_() for t in (Q, K, V))
    ref = ((Q @ K.transpose(-1, -2)) * d ** -0.5).softmax(-1) @ V
    do = rnd((nb * Nq, C), dev, 4)
    gq, gk, gv = torch.autograd.grad(ref, (Q, K, V), _heads(do, nb, Nq, heads, d))
    dqkv = torch.zeros_like(qkv)
    ops.attn_bwd(q, k, v, o, do, lse, Nq, Nq, heads, d, nb, dq=dqkv[:, :C], dk=dqkv[:, C:2 * C], dv=dqkv[:, 2 * C:])
    close(_heads(dqkv[:, :C], nb, Nq, heads, d), gq, 1e-2, 2e-2, "attn dq, 4 tiles per wave")
    close(_heads(dqkv[:, C:2 * C], nb, Nq, heads, d), gk, 1e-2, 2e-2, "attn dk, 4 tiles per wave")
    close(_heads(dqkv[:, 2 * C:], nb, Nq, heads, d), gv, 1e-2, 2e-2, "attn dv, 4 tiles per wave")


@pytest.mark.parametrize("d,long_", [(40, False), (80, False), (40, True)], ids=["d40", "d80", "d40-4-tiles-per-wave"])
def test_attention_rows_that_outgrow_their_first_tile(backend, d, long_):
    """The ring forward keeps a per-row softmax OFFSET (not the running maximum) and re-bases it only when a score exceeds it by
    more than 2^8 (attention.hip: kRebase).  On N(0, 1) operands that happens at the first key tile and almost never again, so
    the re-base of a row that already HAS accumulated output was not exercised - and on the MI355X it was wrong: the rescale
    factor comes out of v_exp_f32 and the first inline-asm v_mul read it one cycle too early (gfx940+ trans forwarding hazard,
    mc_common.hpp: scale_in_place), corrupting element 0 of the first output tile (channels 0 / 4 / 8 / 12 of a head) by the
    row's offset.  Here the scores of half of the rows RAMP UP along the keys by ~45 log2 units (five or six re-bases per row),
    the other half ramp down; forward, lse and the backward (which reads the lse) against fp32 torch, and bit-stable.
    The host simulator has no such hazard: it checks the re-base arithmetic; the GPU run is the regression test."""
    dev = backend
    Nq, heads, nb = (1024, 1, 1) if not big(dev) else (1024 + 128, 4, 3)
    if long_:                      # Nq >= 2048 at d = 40: the 256-row workgroup, four query tiles per wave (level 0 at 512 x 512)
        Nq, heads, nb = (2048, 1, 1) if not big(dev) else (2048 + 64, 4, 2)
    C = heads * d
    g = torch.Generator().manual_seed(21)
    u = torch.randn(heads, d, generator=g)
    u = u / u.norm(dim=1, keepdim=True)
    ramp = torch.linspace(0.0, 1.0, Nq)
    amp = 45.0 / (1.4427 * d ** -0.5)                         # b = 1 rows gain 45 log2 units from the first to the last key
    b = torch.rand(nb, Nq, heads, generator=g) * 2 - 1        # per query row: how strongly (and in which direction) it ramps
    q = 0.7 * torch.randn(nb, Nq, heads, d, generator=g) + b[..., None] * u
    k = 0.7 * torch.randn(nb, Nq, heads, d, generator=g) + (amp * ramp)[None, :, None, None] * u
    v = torch.randn(nb, Nq, heads, d, generator=g)
    q16, k16, v16 = (t.reshape(nb * Nq, C).half().to(dev) for t in (q, k, v))
    o, lse = ops.attn_fwd(q16, k16, v16, Nq, Nq, heads, d, nb)
    o2, lse2 = ops.attn_fwd(q16, k16, v16, Nq, Nq, heads, d, nb)
    assert torch.equal(o, o2) and torch.equal(lse, lse2), "the re-based rows are not bit-stable"
    Q, K, V = (_heads(t, nb, Nq, heads, d).requires_grad_() for t in (q16, k16, v16))
    S = (Q @ K.transpose(-1, -2)) * d ** -0.5
    tile_max = (S.detach() * 1.4427).reshape(nb, heads, Nq, -1, 64).amax(-1) if Nq % 64 == 0 else None
    if tile_max is not None:      # the case is what it claims to be: many rows pass their FIRST tile's maximum by > 8 later on
        assert ((tile_max[..., 1:].amax(-1) - tile_max[..., 0]) > 8).float().mean() > 0.3
    ref = S.softmax(-1) @ V
    close(_heads(o, nb, Nq, heads, d), ref, 1e-2, 1e-2, "attn fwd with re-bases")
    close(lse, torch.logsumexp(S, -1), 2e-3, 2e-3, "attn lse with re-bases")
    do = rnd((nb * Nq, C), dev, 22)
    gq, gk, gv = torch.autograd.grad(ref, (Q, K, V), _heads(do, nb, Nq, heads, d))
    dq, dk, dv = ops.attn_bwd(q16, k16, v16, o, do, lse, Nq, Nq, heads, d, nb)
    close(_heads(dq, nb, Nq, heads, d), gq, 2e-2, 2e-2, "attn dq with re-bases")
    close(_heads(dk, nb, Nq, heads, d), gk, 2e-2, 2e-2, "attn dk with re-bases")
    close(_heads(dv, nb, Nq, heads, d), gv, 2e-2, 2e-2, "attn dv with re-bases")


@pytest.mark.gpu
def test_attention_rebase_negative_control(gpu_device):
    """The regression case above on the SHIPPED library (must be clean) and on the negative control - attention.hip built without
    the s_nop in scale_in_place (build.build_trans_hazard_control) - whose wrong elements are counted and reported
    (gpurun_out/attn_trans_hazard_r05.json): it shows that the ramped scores actually reach the hazard on this GPU."""
    import ctypes
    import json
    import os
    from motionclone_amd import build, lib
    dev = gpu_device
    d, Nq, heads, nb = 40, 1024 + 128, 4, 3
    C = heads * d
    g = torch.Generator().manual_seed(21)
    u = torch.randn(heads, d, generator=g)
    u = u / u.norm(dim=1, keepdim=True)
    amp = 45.0 / (1.4427 * d ** -0.5)
    b = torch.rand(nb, Nq, heads, generator=g) * 2 - 1
    q = 0.7 * torch.randn(nb, Nq, heads, d, generator=g) + b[..., None] * u
    k = 0.7 * torch.randn(nb, Nq, heads, d, generator=g) + (amp * torch.linspace(0.0, 1.0, Nq))[None, :, None, None] * u
    v = torch.randn(nb, Nq, heads, d, generator=g)
    q16, k16, v16 = (t.reshape(nb * Nq, C).half().to(dev) for t in (q, k, v))
    Q, K, V = (_heads(t, nb, Nq, heads, d) for t in (q16, k16, v16))
    ref = ((Q @ K.transpose(-1, -2)) * d ** -0.5).softmax(-1) @ V

    def wrong(out):
        e = (_heads(out, nb, Nq, heads, d) - ref).abs()
        return int((e > 1e-2 + 1e-2 * ref.abs()).sum())
    rows = []
    o, _ = ops.attn_fwd(q16, k16, v16, Nq, Nq, heads, d, nb)
    rows.append(dict(lib="shipped", wrong_elements=wrong(o), elements=o.numel()))
    assert rows[0]["wrong_elements"] == 0
    if os.path.exists(build.TRANS_HAZARD_CONTROL_LIB):
        ctl = ctypes.CDLL(build.TRANS_HAZARD_CONTROL_LIB)
        ctl.mc_attn_fwd_f16.argtypes = lib.SIGNATURES["mc_attn_fwd_f16"]
        ctl.mc_attn_fwd_f16.restype = ctypes.c_int
        worst = 0
        for _ in range(5):
            oc = torch.empty_like(o)
            rc = ctl.mc_attn_fwd_f16(q16.data_ptr(), k16.data_ptr(), v16.data_ptr(), q16.stride(0), k16.stride(0), v16.stride(0),
                                     oc.data_ptr(), oc.stride(0), None, Nq, Nq, heads, d, nb, 1, float(d ** -0.5),
                                     torch.cuda.current_stream().cuda_stream)
            assert rc == 0
            torch.cuda.synchronize()
            worst = max(worst, wrong(oc))
        bad = (_heads(oc, nb, Nq, heads, d) - ref).abs() > 1e-2 + 1e-2 * ref.abs()
        chans = sorted(set(bad.nonzero()[:, 3].tolist()))
        rows.append(dict(lib="control without the s_nop", wrong_elements_worst_of_5=worst, elements=o.numel(), wrong_head_dims=chans))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "attn_trans_hazard_r06.json"), "w") as f:
        json.dump(rows, f, indent=1)
    print("TRANS_HAZARD", json.dumps(rows))
    if len(rows) < 2:
        pytest.skip("shipped library clean; negative-control library %s not built (build.build_trans_hazard_control)"
                    % build.TRANS_HAZARD_CONTROL_LIB)
    # the control must actually go wrong, or this test says nothing about the sensitivity of the regression case above
    assert rows[1]["wrong_elements_worst_of_5"] > 0, "the control (no s_nop) no longer reproduces the TRANS hazard: the regression input is blind"


def test_attention_xcd_block_mapping_is_the_same_arithmetic(backend, monkeypatch):
    """attention.hip attn_block: the 1-D launch that keeps all row blocks of one (batch, head) on one XCD only renames
    workgroups, so forward and backward are bit-identical to the plain (row block, head, batch) grid - also when the number
    of (batch, head) units is not a multiple of the 8 XCDs.  On the emulator MC_ATTN_XCD is re-read per call; on the GPU the
    default already takes the mapping at this size and the comparison is against the oracle-checked tests above."""
    dev = backend
    if big(dev):
        d, Nq, heads, nb = 40, 1024 + 19, 3, 3
    else:
        d, Nq, heads, nb = 16, 150, 3, 3
    C = heads * d
    qkv = rnd((nb * Nq, 3 * C), dev, 11, 0.7)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    do = rnd((nb * Nq, C), dev, 12)

    def run():
        o, lse = ops.attn_fwd(q, k, v, Nq, Nq, heads, d, nb)
        dqkv = torch.zeros_like(qkv)
        ops.attn_bwd(q, k, v, o, do, lse, Nq, Nq, heads, d, nb, dq=dqkv[:, :C], dk=dqkv[:, C:2 * C], dv=dqkv[:, 2 * C:])
        return o, lse, dqkv

    monkeypatch.setenv("MC_ATTN_XCD", "0")
    plain = run()
    monkeypatch.setenv("MC_ATTN_XCD", "2")
    mapped = run()
    for a, b, what in zip(plain, mapped, ("o", "lse", "dqkv")):
        assert torch.equal(a, b), "attention %s differs between the plain grid and the XCD mapping" % what
    Q, K, V = (_heads(t, nb, Nq, heads, d) for t in (q, k, v))
    close(_heads(mapped[0], nb, Nq, heads, d), ((Q @ K.transpose(-1, -2)) * d ** -0.5).softmax(-1) @ V, 1e-2, 1e-2, "attn fwd, XCD mapping")


@pytest.mark.parametrize("d", [40, 80])
@pytest.mark.parametrize("Nq,Nk,share", [(150, 150, 1), (70, 64, 1), (300, 77, 3), (64, 333, 1)])
def test_attention_forward_ring_kernel(backend, monkeypatch, Nq, Nk, share, d):
    """attention.hip attn_fwd_ring_kernel (d = 40 / 80 long sequences: LDS-DMA ring, transpose reads, softmax offset in a k-slot):
    one / several / ragged key tiles, ragged query blocks, keys shared by `share` batch entries.  On the emulator the kernel
    is forced at these small sizes (MC_ATTN_RING is re-read per call) and compared with the register-staged kernel as well;
    on the GPU the sizes are the ones that select it by default."""
    dev = backend
    heads, nb = 2, 2 * share
    if big(dev):
        # (>= 1024 query rows: the sizes that select the ring kernel by default; MC_ATTN_RING is read once per process there)
        Nq, Nk, heads = Nq * (8 if Nq >= 128 else 16) + 5, (Nk * 8 + 3 if Nk != 77 else 77 * 8), 8
    C = heads * d
    q = rnd((nb * Nq, C), dev, 21, 0.7)
    kv = rnd((nb // share * Nk, 2 * C), dev, 22, 0.7)
    k, v = kv[:, :C], kv[:, C:]
    monkeypatch.setenv("MC_ATTN_RING", "2")
    o, lse = ops.attn_fwd(q, k, v, Nq, Nk, heads, d, nb, kv_bdiv=share)
    assert lib.load().mc_attn_last_kernel() == 1, "the ring kernel was not selected"
    Q = _heads(q, nb, Nq, heads, d)
    K = _heads(k, nb // share, Nk, heads, d).repeat_interleave(share, 0)
    V = _heads(v, nb // share, Nk, heads, d).repeat_interleave(share, 0)
    S = (Q @ K.transpose(-1, -2)) * d ** -0.5
    close(_heads(o, nb, Nq, heads, d), S.softmax(-1) @ V, 1e-2, 1e-2, "ring attn fwd")
    close(lse, torch.logsumexp(S, -1), 2e-3, 1e-3, "ring attn lse")
    if not big(dev):
        monkeypatch.setenv("MC_ATTN_RING", "0")
        o0, lse0 = ops.attn_fwd(q, k, v, Nq, Nk, heads, d, nb, kv_bdiv=share)
        close(o, o0, 2e-3, 2e-3, "ring vs register-staged forward")
        close(lse, lse0, 1e-4, 1e-4, "ring vs register-staged lse")


@pytest.mark.parametrize("d", [40, 80])
@pytest.mark.parametrize("Nq,Nk,share", [(148, 148, 1), (64, 64, 1), (300, 77, 3)])
def test_attention_backward_ring_kernels(backend, monkeypatch, Nq, Nk, share, d):
    """attention.hip attn_bwd_dq_ring_kernel / attn_bwd_dkdv_ring_kernel (d = 40 / 80 long sequences): lse and D carried in padding
    k-slots of the dQ kernel's MFMAs, Q / dO / lse / D streamed by LDS-DMA in the dK / dV kernel.  Against autograd; ragged
    tiles on both axes; dq only with shared keys (cross-attention).  Forced at these sizes on the emulator, default on the GPU."""
    dev = backend
    heads, nb = 2, 2 * share
    if big(dev):
        Nq, Nk, heads = Nq * (8 if Nq >= 128 else 16) + 4, (Nk * 8 + 4 if Nk != 77 else 77 * 8), 8
        if share == 1:
            Nk = Nq
    C = heads * d
    q = rnd((nb * Nq, C), dev, 31, 0.7)
    kv = rnd((nb // share * Nk, 2 * C), dev, 32, 0.7)
    k, v = kv[:, :C], kv[:, C:]
    do = rnd((nb * Nq, C), dev, 33)
    monkeypatch.setenv("MC_ATTN_RING", "2")
    o, lse = ops.attn_fwd(q, k, v, Nq, Nk, heads, d, nb, kv_bdiv=share)
    Q = _heads(q, nb, Nq, heads, d).requires_grad_()
    K0 = _heads(k, nb // share, Nk, heads, d).requires_grad_()
    V0 = _heads(v, nb // share, Nk, heads, d).requires_grad_()
    K, V = K0.repeat_interleave(share, 0), V0.repeat_interleave(share, 0)
    ref = ((Q @ K.transpose(-1, -2)) * d ** -0.5).softmax(-1) @ V
    gq, gk, gv = torch.autograd.grad(ref, (Q, K0, V0), _heads(do, nb, Nq, heads, d))
    if share == 1:
        dq, dk, dv = ops.attn_bwd(q, k, v, o, do, lse, Nq, Nk, heads, d, nb)
        assert lib.load().mc_attn_last_kernel() == 3, "the ring backward kernels were not selected"
        close(_heads(dk, nb, Nk, heads, d), gk, 1e-2, 2e-2, "ring attn dk")
        close(_heads(dv, nb, Nk, heads, d), gv, 1e-2, 2e-2, "ring attn dv")
    else:
        dq, _, _ = ops.attn_bwd(q, k, v, o, do, lse, Nq, Nk, heads, d, nb, kv_bdiv=share, need_dkv=False)
        assert lib.load().mc_attn_last_kernel() == 1
    close(_heads(dq, nb, Nq, heads, d), gq, 1e-2, 2e-2, "ring attn dq")
    if not big(dev):   # and against the register-staged kernels
        monkeypatch.setenv("MC_ATTN_RING", "0")
        if share == 1:
            dq0, dk0, dv0 = ops.attn_bwd(q, k, v, o, do, lse, Nq, Nk, heads, d, nb)
            close(dk, dk0, 3e-3, 3e-3, "ring vs register-staged dk")
            close(dv, dv0, 3e-3, 3e-3, "ring vs register-staged dv")
        else:
            dq0, _, _ = ops.attn_bwd(q, k, v, o, do, lse, Nq, Nk, heads, d, nb, kv_bdiv=share, need_dkv=False)
        close(dq, dq0, 3e-3, 3e-3, "ring vs register-staged dq")


def test_cross_attention_fwd_bwd(backend):
    dev = backend
    heads, d, B, F_, N, Nk = 2, 40, 2, 3, 50, 77
    if big(dev):
        heads, d, N = 8, 40, 1024
    C = heads * d
    nb = B * F_
    q = rnd((nb * N, C), dev, 1, 0.7)
    kv = rnd((B * Nk, 2 * C), dev, 2, 0.7)
    k, v = kv[:, :C], kv[:, C:]
    o, lse = ops.attn_fwd(q, k, v, N, Nk, heads, d, nb, kv_bdiv=F_)
    Q = _heads(q, nb, N, heads, d).requires_grad_()
    K = _heads(k, B, Nk, heads, d).repeat_interleave(F_, 0)
    V = _heads(v, B, Nk, heads, d).repeat_interleave(F_, 0)
    ref = ((Q @ K.transpose(-1, -2)) * d ** -0.5).softmax(-1) @ V
    close(_heads(o, nb, N, heads, d), ref, 1e-2, 1e-2, "xattn fwd")
    do = rnd((nb * N, C), dev, 3)
    (gq,) = torch.autograd.grad(ref, Q, _heads(do, nb, N, heads, d))
    dq, _, _ = ops.attn_bwd(q, k, v, o, do, lse, N, Nk, heads, d, nb, kv_bdiv=F_, need_dkv=False)
    close(_heads(dq, nb, N, heads, d), gq, 1e-2, 2e-2, "xattn dq")


# ---------------------------------------------------------------------------------------------------
def _temporal_ref(qkv, B, F_, HW, heads, d):
    C = heads * d
    t = qkv.float().reshape(B, F_, HW, 3, heads, d).permute(3, 0, 2, 4, 1, 5)  # [3, B, HW, heads, F, d]
    return t[0].reshape(-1, heads, F_, d), t[1].reshape(-1, heads, F_, d), t[2].reshape(-1, heads, F_, d)


def _temporal_unref(t, B, F_, HW, heads, d):  # [B*HW, heads, F, d] -> [(b f hw), heads*d]
    return t.reshape(B, HW, heads, F_, d).permute(0, 3, 1, 2, 4).reshape(B * F_ * HW, heads * d)


@pytest.mark.parametrize("F_,d", [(16, 40), (5, 16), (16, 160), (24, 32), (32, 80), (32, 160), (32, 40)])
def test_temporal_attention_and_guidance(backend, F_, d):
    dev = backend
    B, HW, heads = (2, 6, 2) if not big(dev) else (2, 300, 8)
    C = heads * d
    qkv = rnd((B * F_ * HW, 3 * C), dev, 1, 0.8)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    o = ops.tattn_fwd(q, k, v, B, F_, HW, heads, d)
    # head dims 40 / 80 / 160 with 16-byte readable rows take the vector-load kernel (temporal.hip tattn_fwd_vec_kernel)
    assert lib.load().mc_tattn_last_kernel() == (1 if d in (40, 80, 160) else 0)
    Q, K, V = (t.requires_grad_() for t in _temporal_ref(qkv, B, F_, HW, heads, d))
    P = ((Q @ K.transpose(-1, -2)) * d ** -0.5).softmax(-1)  # [B*HW, heads, F, F]
    ref = P @ V
    close(o, _temporal_unref(ref, B, F_, HW, heads, d), 1e-2, 1e-2, "tattn fwd")

    close(ops.tattn_prob(q, k, B, F_, HW, heads, d), P, 2e-3, 2e-3, "tattn prob")
    # extraction: top-1 of P (motionclone_functions.py:79)
    val, idx = ops.tattn_top1(q, k, B, F_, HW, heads, d)
    rv, ri = torch.topk(P, 1, -1)
    close(val, rv, 2e-3, 2e-3, "top1 value")
    mism = (idx.long().cpu() != ri.cpu())
    if mism.any():  # only acceptable at numerical ties
        p2 = torch.gather(P, -1, idx.long().to(P.device))
        assert ((rv - p2).abs()[mism.to(rv.device)] < 1e-3).all(), "top1 index mismatch beyond a tie"

    # loss + gradient with a perturbed reference representation
    ref_idx = torch.randint(0, F_, ri.shape, generator=torch.Generator().manual_seed(7)).to(torch.uint8).to(dev)
    ref_val = (torch.rand(ri.shape, generator=torch.Generator().manual_seed(8)) * 0.5).to(dev)
    loss = ops.tattn_loss(q, k, ref_idx, ref_val, B, F_, HW, heads, d)
    gathered = torch.gather(P, -1, ref_idx.long())
    loss_ref = Fn.mse_loss(gathered, ref_val)
    assert abs(loss.item() - loss_ref.item()) < 2e-3 * max(1.0, abs(loss_ref.item())) + 1e-5

    weight = 300.0
    do = rnd((B * F_ * HW, C), dev, 3)
    dO = _temporal_ref(torch.cat([do, do, do], 1), B, F_, HW, heads, d)[0]
    total = (ref * dO).sum() + weight * loss_ref
    gq, gk, gv = torch.autograd.grad(total, (Q, K, V), retain_graph=True)
    dqkv = torch.zeros_like(qkv)
    coef = weight * 2.0 / gathered.numel()
    ops.tattn_bwd(q, k, v, do, dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:], B, F_, HW, heads, d,
                  ref_idx=ref_idx, ref_val=ref_val, seed_coef=coef)
    close(dqkv[:, :C], _temporal_unref(gq, B, F_, HW, heads, d), 1e-2, 2e-2, "tattn dq")
    close(dqkv[:, C:2 * C], _temporal_unref(gk, B, F_, HW, heads, d), 1e-2, 2e-2, "tattn dk")
    close(dqkv[:, 2 * C:], _temporal_unref(gv, B, F_, HW, heads, d), 1e-2, 2e-2, "tattn dv")

    # seed only (dO = NULL): what the last hooked attention of up_blocks.1 sees
    gq2, gk2 = torch.autograd.grad(weight * Fn.mse_loss(torch.gather(P, -1, ref_idx.long()), ref_val), (Q, K))
    d2 = torch.ones_like(qkv)
    ops.tattn_bwd(q, k, v, None, d2[:, :C], d2[:, C:2 * C], d2[:, 2 * C:], B, F_, HW, heads, d,
                  ref_idx=ref_idx, ref_val=ref_val, seed_coef=coef)
    close(d2[:, :C], _temporal_unref(gq2, B, F_, HW, heads, d), 2e-3, 2e-2, "seed dq")
    close(d2[:, C:2 * C], _temporal_unref(gk2, B, F_, HW, heads, d), 2e-3, 2e-2, "seed dk")
    assert d2[:, 2 * C:].abs().max() == 0


# ---------------------------------------------------------------------------------------------------
def test_elementwise(backend):
    dev = backend
    M, D = (37, 64) if not big(dev) else (4099, 1280)
    x = rnd((M, 2 * D), dev, 1)
    xr = x.float().requires_grad_()
    ref = xr[:, :D] * Fn.gelu(xr[:, D:])
    close(ops.geglu_fwd(x), ref, 1e-2, 5e-3, "geglu")
    dy = rnd((M, D), dev, 2)
    (g,) = torch.autograd.grad(ref, xr, dy.float())
    close(ops.geglu_bwd(dy, x), g, 1e-2, 1e-2, "geglu bwd")
    a, b = rnd((M, D), dev, 3), rnd((M, D), dev, 4)
    close(ops.add(a, b, sa=0.5, sb=2.0), 0.5 * a.float() + 2 * b.float(), 1e-2, 2e-3, "add")
    close(ops.silu(a), Fn.silu(a.float()), 1e-3, 2e-3, "silu")
    NF, H, W, C = 2, 3, 5, 64
    up = rnd((NF, C, 2 * H, 2 * W), dev, 5)
    close(_from_cl(ops.sumpool2(_to_cl(up), NF, H, W), NF, H, W), Fn.avg_pool2d(up.float(), 2) * 4, 1e-2, 2e-3, "sumpool")
    lat = rnd((2, 4, 3, 5, 6), dev, 6)
    cl = ops.latent_to_cl(lat, 64)
    assert cl.shape == (2 * 3 * 30, 64) and cl[:, 4:].abs().max() == 0
    back = ops.cl_to_latent(cl, 2, 4, 3, 5, 6)
    assert torch.equal(back, lat)
    t = torch.tensor([400.0, 37.0], device=dev)
    emb = ops.timestep_embed(t, 320, lat)
    freqs = torch.exp(-math.log(10000.0) * torch.arange(160, device=dev).float() / 160)
    e = t[:, None] * freqs[None]
    close(emb, torch.cat([torch.cos(e), torch.sin(e)], -1), 2e-3, 2e-3, "timestep embedding")


def test_cfg_ddim_step(backend):
    dev = backend
    CL, F_, H, W = 4, 3, 5, 6
    x = rnd((1, CL, F_, H, W), dev, 1)
    ec = rnd((1, CL, F_, H, W), dev, 2)
    eu = rnd((1, CL, F_, H, W), dev, 3)
    score = torch.randn(1, CL, F_, H, W, generator=torch.Generator().manual_seed(4)).to(dev)
    ec_cl, eu_cl = ops.latent_to_cl(ec, 64), ops.latent_to_cl(eu, 64)
    a_t, a_prev, cfg, gs = 0.31, 0.42, 7.5, 1.0
    out = ops.cfg_ddim_step(ec_cl, eu_cl, x, score, cfg, a_t, a_prev, gs * math.sqrt(1 - a_t))
    eps = ec.float() + cfg * (ec.float() - eu.float())
    x0 = (x.float() - math.sqrt(1 - a_t) * eps) / math.sqrt(a_t)
    eps2 = eps - gs * math.sqrt(1 - a_t) * score
    ref = math.sqrt(a_prev) * x0 + math.sqrt(1 - a_prev) * eps2
    close(out, ref, 2e-2, 3e-3, "cfg+ddim guided")
    out2 = ops.cfg_ddim_step(ec_cl, eu_cl, x, None, cfg, a_t, a_prev, 0.0)
    close(out2, math.sqrt(a_prev) * x0 + math.sqrt(1 - a_prev) * eps, 2e-2, 3e-3, "cfg+ddim plain")


@pytest.mark.parametrize("sizes", [(37, 53, 32, 48), (90, 160, 64, 64), (48, 48, 64, 96), (123, 77, 50, 91), (20, 24, 16, 16)])
def test_video_resize(backend, sizes):
    """reference-video front end (util.py:232-238): F.interpolate(bilinear, align_corners=True) ON THE uint8 FRAMES, then
    / 127.5 - 1.  Byte work: the kernel's default mode reproduces torch's CPU uint8 result EXACTLY - the resized level of
    every element, recovered from the fp16 output, equals torch's uint8 value (down- and up-scaling, odd sizes)."""
    dev = backend
    hs, ws, H, W = sizes
    g = torch.Generator().manual_seed(hs * 1000 + ws)
    x = torch.randint(0, 256, (3, hs, ws, 3), generator=g, dtype=torch.uint8)
    got = ops.video_resize(x.to(dev), H, W)
    assert got.shape == (3, 3, H, W) and got.dtype == torch.float16
    u8 = Fn.interpolate(x.permute(0, 3, 1, 2), size=(H, W), mode="bilinear", align_corners=True)   # the reference's call
    assert u8.dtype == torch.uint8
    want = (u8.float() / 127.5 - 1.0).half()                 # video / 127.5 - 1.0, stored as the fp16 the VAE consumes
    assert torch.equal(got.cpu(), want), "resize differs from torch's uint8 path in %d elements" % int((got.cpu() != want).sum())
    levels = torch.round((got.float().cpu() + 1.0) * 127.5).to(torch.uint8)
    assert torch.equal(levels, u8)
    # the float modes: un-quantised, rounded, truncated
    xf = x.permute(0, 3, 1, 2).float()
    ref = Fn.interpolate(xf, size=(H, W), mode="bilinear", align_corners=True)
    smooth = ops.video_resize(x.to(dev), H, W, quantise=0)
    assert (smooth.float().cpu() - (ref / 127.5 - 1.0)).abs().max() < 2e-3
    for q, fn in ((1, torch.round), (3, torch.floor)):
        alt = ops.video_resize(x.to(dev), H, W, quantise=q).float().cpu()
        assert ((alt - (fn(ref) / 127.5 - 1.0)).abs() > 2e-3).float().mean() < 0.01      # float ties at .0 / .5 boundaries only
    from motionclone_amd.utils.util import pick_frames, preprocess_frames
    assert pick_frames(100, 4).tolist() == [0, 33, 66, 99] and pick_frames(100, 3, fps=10.0, duration=2.05).tolist() == [0, 9, 19]
    assert torch.equal(preprocess_frames(x.numpy(), H, W, device=dev), got)


@pytest.mark.parametrize("F_,d", [(16, 160), (16, 40), (24, 32), (32, 160), (32, 40)])
def test_top1_follows_the_reference_fp16_order_bit_exactly(backend, F_, d):
    """obtain_motion_representation in the reference's own arithmetic (attention.py:593-609, motionclone_functions.py:79):
    scores rounded to fp16 -> fp32 softmax -> probabilities rounded to fp16 -> topk(k=1) on the fp16 values.  Checked
    bit for bit (uint8 indices AND fp16 values, and the full fp16 probability tensor of get_temp_attn_prob) against an
    exact-arithmetic (fp64) emulation of that order.  A row may differ only if it holds a value within the fp32 dot-product
    error (1e-6 of sum |q_i k_i|) of an fp16 rounding boundary - where the reference's own result depends on its GEMM's summation order - and fewer than 2 % do (0.00x % at full size).
    Ties between equal fp16 probabilities go to the lowest index; a duplicated key frame makes exact ties certain."""
    dev = backend
    B, HW, heads = (1, 12, 2) if not big(dev) else (2, 700, 8)
    C = heads * d
    qkv = rnd((B * F_ * HW, 3 * C), dev, 11, 0.9)
    # exact ties: key frame 3 := key frame 1 for every unit
    t = qkv.view(B, F_, HW, 3 * C)
    t[:, 3, :, C:2 * C] = t[:, 1, :, C:2 * C]
    q, k = qkv[:, :C], qkv[:, C:2 * C]
    val, idx = ops.tattn_top1(q, k, B, F_, HW, heads, d)
    prob = ops.tattn_prob(q, k, B, F_, HW, heads, d)
    Q, K, _ = _temporal_ref(qkv.cpu(), B, F_, HW, heads, d)
    scale = float(torch.tensor(d ** -0.5, dtype=torch.float32))
    s64 = (Q.double() @ K.double().transpose(-1, -2)) * scale                      # [B*HW, heads, F, F]
    eps = 1e-6

    def near_boundary(x64, tol):
        return (x64 + tol).half() != (x64 - tol).half()
    # the kernel's (and the reference GEMM's) fp32 dot product carries an ABSOLUTE error ~ 2^-24 * sum |q_i k_i|, which near a
    # cancelling score is far more than 1e-6 of the score itself
    mag = (Q.double().abs() @ K.double().abs().transpose(-1, -2)) * scale
    s16 = s64.half()
    p64 = torch.softmax(s16.double(), dim=-1)
    p16 = p64.half()
    amb = (near_boundary(s64, eps * mag) | near_boundary(p64, eps * p64)).any(-1)  # [B*HW, heads, F]
    want_val = p16.max(-1, keepdim=True).values
    want_idx = (p16 == want_val).int().argmax(-1, keepdim=True)                    # first (lowest) index among ties
    bad = (idx.cpu()[..., 0].long() != want_idx[..., 0]) | (val.cpu()[..., 0] != want_val[..., 0]) \
        | (prob.cpu() != p16).any(-1)
    # every row that is not bit-identical must hold a value on an fp16 rounding boundary, and such rows must be rare
    assert not (bad & ~amb).any(), "%d rows differ from the reference order away from any rounding boundary" % int((bad & ~amb).sum())
    assert bad.float().mean().item() < 0.02, bad.float().mean().item()
    tie_rows = ((p16 == want_val).sum(-1) > 1) & ~bad
    assert tie_rows.any(), "the duplicated key frame should have produced exact ties"
    print("top1 bit-exact on %d of %d rows (%d differ, all on an fp16 rounding boundary; %d rows near one), %d exact with ties"
          % (int((~bad).sum()), bad.numel(), int(bad.sum()), int(amb.sum()), int(tie_rows.sum())))


@pytest.mark.parametrize("kind,N,geglu,with_pe", [(1, 320, False, False), (1, 960, False, True), (1, 640, True, False),
                                                  (2, 320, False, False), (2, 128, True, False)])
def test_norm_gemm_equals_norm_then_gemm(backend, kind, N, geglu, with_pe):
    """mc_norm_gemm_f16 (round 4): LayerNorm (+ position table) / GroupNorm-without-activation applied to the rows of A in
    registers inside the K = 320 streaming GEMM, against the two-launch form it replaces (same arithmetic, one fp16 rounding
    of the normalised value; the statistics agree to fp32 rounding) and against fp32 torch.  M = 3 frames x 256 tokens with a
    ragged last workgroup for LayerNorm."""
    dev = backend
    K, hw = 320, 256
    M = 3 * hw if (kind == 2 or with_pe) else 3 * hw - 40
    x = (rnd((M, K), dev, 1) * 1.5 + 0.3).half()
    w = (rnd((N, K), dev, 2) * 0.05).half()
    bias = rnd((1, N), dev, 3).float() * 0.1
    gamma, beta = 1.0 + 0.2 * rnd((K,), dev, 4).float(), 0.1 * rnd((K,), dev, 5).float()
    pe = rnd((4, K), dev, 6).float() if with_pe else None
    if geglu:
        wi, bi = ops.interleave_geglu(w).contiguous(), ops.interleave_geglu(bias.reshape(-1)).reshape(1, -1).contiguous()
    else:
        wi, bi = w, bias
    got = ops.norm_gemm(x, wi, kind, gamma, beta, bias=bi, pe=pe, hw=hw, eps=1e-5, geglu=geglu, force=True)
    assert got is not None, "the fused kernel refused a shape it is specified for"
    out, stats = got
    if kind == 1:
        n, ls = ops.layernorm_fwd(x, gamma, beta, eps=1e-5, pe=pe, hw=hw)
        xf = x.float()
        ref_n = torch.nn.functional.layer_norm(xf, (K,), gamma, beta, 1e-5)
        if pe is not None:
            ref_n = ref_n + pe[(torch.arange(M, device=dev) // hw) % 4]
    else:
        n, ls = ops.gn_fwd(x, None, gamma, beta, False, M // hw, hw, 1e-5)
        xr = x.float().reshape(M // hw, hw, K).permute(0, 2, 1)
        ref_n = torch.nn.functional.group_norm(xr, 32, gamma, beta, 1e-5).permute(0, 2, 1).reshape(M, K)
    two = ops.gemm(n, wi, bias=bi, geglu=geglu)
    close(stats, ls, 1e-4, 1e-4, "statistics vs the separate norm kernel")
    close(out, two, 4e-3, 4e-3, "fused vs norm kernel + GEMM")
    y = ref_n.half().float() @ w.float().t() + bias
    if geglu:
        y = y[:, :N // 2] * torch.nn.functional.gelu(y[:, N // 2:])
    close(out, y, 1e-2, 1e-2, "fused vs fp32 torch")
    assert ops.norm_gemm(x[:, :256].contiguous(), wi[:, :256].contiguous(), kind, gamma[:256], beta[:256], force=True) is None


def test_gemm5_two_workgroups_per_cu_geometry(backend):
    """gemm5 with 256 x 160 tiles, 4 waves, ring of three stages (cfg = 9, round 4): two workgroups per CU out of step.  Dense
    with per-batch bias / residual / alpha, M and N tails, K of 2, 4, 6 stages (ring not yet full) and 20 / 40, fused GEGLU;
    the same sums in the same order as the 8-wave kernel (a wave tile is 64 x 160 in both), so the two agree bit for bit."""
    dev = backend
    shapes = [(300, 320, 64), (260, 640, 128), (520, 480, 192), (300, 160, 640)] if not big(dev) else \
        [(3000, 960, 64), (5000, 640, 1280), (32768, 640, 640), (4111, 1120, 192)]
    for (M, N, K) in shapes:
        a, w = rnd((M, K), dev, 1), rnd((N, K), dev, 2, 0.1)
        rpb = 256 if M > 256 else M          # (a tile reads one bias row: rows_per_batch is a multiple of the tile height)
        bias = torch.randn((M + rpb - 1) // rpb, N, generator=torch.Generator().manual_seed(3)).to(dev)
        res = rnd((M, N), dev, 4)
        out = ops.gemm(a, w, bias=bias, residual=res, alpha=0.5, rows_per_batch=rpb, cfg=9)
        lin = (0.5 * (a.float() @ w.float().t()) + bias.repeat_interleave(rpb, 0)[:M]).half().float()   # rounded, then + R
        close(out, lin + res.float(), 2e-2, 5e-3, "256x160 dense %d %d %d" % (M, N, K))
        assert lib.load().mc_gemm_last_kernel() == 56
        assert torch.equal(out, ops.gemm(a, w, bias=bias, residual=res, alpha=0.5, rows_per_batch=rpb, cfg=11))
    M, K, D = (300, 128, 80) if not big(dev) else (3000, 640, 640)
    a = rnd((M, K), dev, 1)
    wg = rnd((2 * D, K), dev, 8, 0.1)
    bg = torch.randn(1, 2 * D, generator=torch.Generator().manual_seed(9)).to(dev)
    y = a.float() @ wg.float().t() + bg
    og = ops.gemm(a, ops.interleave_geglu(wg), bias=ops.interleave_geglu(bg.t()).t().contiguous(), geglu=True, cfg=9)
    close(og, y[:, :D] * Fn.gelu(y[:, D:]), 2e-2, 1e-2, "256x160 geglu")
    assert torch.equal(og, ops.gemm(a, ops.interleave_geglu(wg), bias=ops.interleave_geglu(bg.t()).t().contiguous(), geglu=True, cfg=11))


def test_norm_gemm_layernorm_statistics_with_a_large_row_mean(backend):
    """the fused LayerNorm takes var = E[x^2] - mean^2 from fp32 sums (the separate kernel subtracts the mean first): rows whose
    mean is 16 standard deviations - far beyond what the UNet's hidden states show - still give rstd to 1e-4 of the exact value"""
    dev = backend
    M, K, N = 256, 320, 64
    g = torch.Generator().manual_seed(12)
    x = (8.0 + 0.5 * torch.randn(M, K, generator=g)).half().to(dev)
    w = rnd((N, K), dev, 2, 0.05)
    gamma, beta = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    out, stats = ops.norm_gemm(x, w, 1, gamma, beta, force=True)
    xd = x.double().cpu()
    mean, var = xd.mean(1), xd.var(1, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    assert ((stats[:, 0].double().cpu() - mean).abs() / mean.abs()).max() < 1e-6
    assert ((stats[:, 1].double().cpu() - rstd).abs() / rstd).max() < 1e-4
    y = ((xd - mean[:, None]) * rstd[:, None]).half().double() @ w.double().cpu().t()
    close(out, y, 2e-2, 1e-2, "fused LayerNorm + GEMM on rows with mean = 16 sigma")
