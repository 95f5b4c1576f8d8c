"""Thin tensor-level wrappers over the C ABI (include/mc_kernels.h).

PyTorch is plumbing here: it owns device memory and the stream; every arithmetic step is a call
into libmotionclone_hip.so.  Activations are fp16 token matrices [tokens, C] (token order
(b, f, y, x)); a strided column window of a wider matrix is passed as a view (stride(1) == 1).
"""
import threading

import torch

from . import lib

DENSE, CONV_S1, CONV_S2, CONV_UP, TCONV_S2 = 0, 1, 2, 3, 4


class _Pinned(threading.local):
    stream = None   # HIP stream handle pinned for the duration of one engine call (see `scoped`); per host thread, like
                    # torch's current stream, so engines driven from several threads keep their own streams
    share = None    # tile-policy override of the engine whose call is running (`gemm_lanes` attribute, see `scoped`)
    owner = None    # the outermost object whose `scoped` call is running: owns the tile-loop counter blocks of its launches


_PIN = _Pinned()


def scoped(fn):
    """Decorator for the engine entry points: look the current HIP stream up once per call instead of once per kernel
    launch (the lookup is ~1.8 us of the ~8 us a wrapper call costs on the host, and a step issues ~1000 launches).
    Evaluated when the method is entered, so a call made under hipGraph capture pins the capturing stream."""
    def wrapper(self, *a, **k):
        pin = _PIN
        prev, prev_share, prev_owner = pin.stream, pin.share, pin.owner
        if prev_owner is None:
            pin.owner = self
        dev = getattr(self, "dev", None)
        if prev is None and dev is not None and dev.type == "cuda":
            pin.stream = torch.cuda.current_stream(dev).cuda_stream
        # an engine / sampler that sets `gemm_lanes` (the launch sequences ITS owner keeps in flight) overrides the process-wide
        # set_gemm_share for the duration of its calls: two samplers with different lane counts can share a process
        lanes = getattr(self, "gemm_lanes", None)
        if lanes is None:
            lanes = getattr(getattr(self, "engine", None), "gemm_lanes", None)
        if lanes is not None:
            pin.share = _share_of(lanes)
        try:
            return fn(self, *a, **k)
        finally:
            pin.stream, pin.share, pin.owner = prev, prev_share, prev_owner
    wrapper.__name__, wrapper.__doc__ = fn.__name__, fn.__doc__
    return wrapper


def _stream(t):
    if t.is_cuda:
        st = _PIN.stream
        return st if st is not None else torch.cuda.current_stream(t.device).cuda_stream
    if not lib.is_emulated():
        raise RuntimeError("motionclone_amd kernels run on an MI355X; got a CPU tensor and no GPU library "
                           "(there is no CPU fallback)")
    return 0


def _p(t):
    return None if t is None else t.data_ptr()


def _ld(t):
    if t is None:
        return 0
    assert t.dim() == 2 and t.stride(1) == 1, "expected a row-major 2-D (possibly column-windowed) matrix"
    return t.stride(0)


def _f16(t):
    assert t.dtype == torch.float16, t.dtype
    return t


def _f32(t):
    assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    return t


def empty(shape, like, dtype=torch.float16):
    return torch.empty(shape, dtype=dtype, device=like.device)


def workspace(op, like, *dims):
    """caller-owned workspace of `op`, sized by the library's own mc_workspace_bytes_<op> (include/mc_kernels.h)"""
    return torch.empty(lib.workspace_bytes(op, *dims) // 4, dtype=torch.float32, device=like.device)


# ---- GEMM family ---------------------------------------------------------------------------------
_SPLIT_PLAN = {}   # (M, N, K, mode) -> mc_gemm_splitk_plan, memoised: one ctypes round trip less per launch
_GEMM_SHARE = 0    # log2 of the launch sequences kept in flight on separate streams (set_gemm_share)


def _share_of(lanes):
    return 0 if lanes <= 1 else (1 if lanes < 4 else 2)


def set_gemm_share(lanes):
    """Tell the GEMM tile / split-K choice how many independent launch sequences share the GPU (sample_interleaved:
    one per video in flight).  A launch then only has to fill 1 / lanes of the CUs (mc_gemm_f16 flags bits 20-21).  The
    choice changes fp32 summation order, not arithmetic: results are reproducible for a given setting.  This is the
    PROCESS-WIDE default; an engine or sampler with a `gemm_lanes` attribute overrides it for its own calls (`scoped`)."""
    global _GEMM_SHARE
    _GEMM_SHARE = _share_of(lanes)
    return _GEMM_SHARE


def gemm_share():
    """the setting in force for the calling thread: the running engine call's override, else the process-wide default"""
    return _GEMM_SHARE if _PIN.share is None else _PIN.share


# ---- persistent tile loop (mc_gemm_tileloop_f16) ------------------------------------------------------------------------
# Counter blocks of the dynamic tile order / stream-K flags: 2 KiB of zero bytes per (owner, stream).  The kernel zeroes its block before it ends, and
# the launches of one owner on one stream (or inside the graphs captured from it) run one after the other, so ONE block per
# key is enough; keys never share a block, so launch sequences that overlap (lanes) never meet in a counter.  Blocks are cut
# from a slab that is allocated and zeroed OUTSIDE any graph capture (handing out a block is pointer arithmetic).
_TILE_SLAB_BLOCKS = 512
_TILE_BLOCK_BYTES = 2048
_tile_slabs = {}     # device -> zeroed int32 tensor
_tile_blocks = {}    # (id(owner), device, stream) -> byte address
_tile_lock = threading.Lock()
TILELOOP_COL_OUTER = False   # tile order of the loop: column blocks outermost (mc_gemm_tileloop_f16 flags 0x4); A/B
TILELOOP = None      # None: the measured policy of `_tileloop_wanted`; False: never; True: wherever the kernel accepts the shape


def prepare_tile_counters(device):
    """allocate the counter slab of `device` (call outside a graph capture; engines do at construction)"""
    key = str(device)
    with _tile_lock:
        if key not in _tile_slabs:
            _tile_slabs[key] = torch.zeros(_TILE_SLAB_BLOCKS * _TILE_BLOCK_BYTES // 4, dtype=torch.int32, device=device)
    return _tile_slabs[key]


def _tile_counter_block(t, stream):
    """address of the calling owner's zeroed counter block for launches on `stream`, or None (= static tile order) when
    the slab cannot be created right now (first use inside a graph capture) or is used up"""
    key = (id(_PIN.owner) if _PIN.owner is not None else 0, str(t.device), stream)
    addr = _tile_blocks.get(key)
    if addr is None:
        dkey = str(t.device)
        slab = _tile_slabs.get(dkey)
        if slab is None:
            if t.is_cuda and torch.cuda.is_current_stream_capturing():
                return None
            slab = prepare_tile_counters(t.device)
        with _tile_lock:
            n = sum(1 for k in _tile_blocks if k[1] == dkey)
            if n >= _TILE_SLAB_BLOCKS:
                return None
            addr = _tile_blocks[key] = slab.data_ptr() + _TILE_BLOCK_BYTES * n
    return addr


def _tileloop_wanted(M, N, K, share):
    """measured policy (profiles/r06_tileloop.md): every DENSE problem that mc_gemm_f16 would run on gemm5's 256x320 tiles - the
    persistent loop is 4 ... 20 % faster on all of them (the K = 320 layers of the 64x64 level stay on the streaming kernel)"""
    if N % 320 or K < 256 or K % 64:
        return False
    if K == 320 and (M >= 98304 or (M >= 32768 and N >= 640)):
        return False
    rows = (M + 255) // 256
    return rows >= 8 and rows * (N // 320) >= (224 >> share)


def gemm_tileloop(a, w, *, a2=None, bias=None, residual=None, out=None, alpha=1.0, rows_per_batch=0, geglu=False,
                  dynamic=True, strict_order=False, max_wg=0, stream_k=False, col_outer=None):
    """`gemm` (DENSE) on the persistent tile loop: out = alpha * [a | a2] . w^T + bias + residual, bit-identical to the
    256x320 kernel of mc_gemm_f16.  Returns None when the shape is outside the kernel (caller: `gemm`)."""
    _f16(a), _f16(w)
    N, K = w.shape
    M, c1 = a.shape
    assert c1 + (a2.shape[1] if a2 is not None else 0) == K
    n_out = N // 2 if geglu else N
    if out is None:
        out = empty((M, n_out), a)
    assert out.shape[0] == M and out.shape[1] == n_out
    if bias is not None:
        _f32(bias)
        assert bias.shape[-1] == N
    st = _stream(a)
    ctr = _tile_counter_block(a, st) if (dynamic or stream_k) else None
    part = None
    if stream_k:
        if ctr is None:
            return None
        part = torch.empty(lib.workspace_bytes("gemm_tileloop", 1) // 4, dtype=torch.float32, device=a.device)
    if col_outer is None:
        col_outer = TILELOOP_COL_OUTER
    flags = (0x200 if geglu else 0) | (1 if strict_order else 0) | (2 if stream_k else 0) | (4 if col_outer else 0) | ((max_wg // 8) << 16)
    ok = lib.try_call("mc_gemm_tileloop_f16", _p(a), _p(a2), _p(w), _p(out), _p(residual), _p(bias), M, N, K, _ld(a),
                      _ld(a2), _ld(out), _ld(residual), c1, rows_per_batch, float(alpha), flags, ctr,
                      _TILE_BLOCK_BYTES if ctr else 0, _p(part), part.numel() * 4 if part is not None else 0, st)
    return out if ok else None


def gemm(a, w, *, a2=None, bias=None, residual=None, out=None, mode=DENSE, geom=None, alpha=1.0,
         rows_per_batch=0, tile=0, m_out=None, geglu=False, deep=False, cfg=0, splits=None,
         pad_front=True, nsplit=0, g3_splitk=False, tileloop=None, gn_hw=0):
    """out[M,N] = alpha * gather(a, a2) . w[N,K]^T + bias + residual.

    gn_hw > 0 (round 6): the caller will GroupNorm `out` next, in frames of gn_hw tokens -> returns (out, gnp) where gnp =
    (partial sums written by the producing kernel's epilogue, chunks per frame) for `gn_fwd(..., gnp=gnp)`, or None when the
    library's choice for this problem leaves no statistics (then gn_fwd runs its own statistics pass).

    geom = (Hs, Ws, Ho, Wo) for the conv modes; m_out = number of output tokens for conv modes.
    geglu=True: w rows interleaved (h_j, gate_j) (see `interleave_geglu`), out gets N/2 columns h*gelu(gate).
    splits: K ranges for the split-K path (None = the library's plan, 1 = off).
    pad_front=False (CONV_S2 only): zero padding only right / bottom, i.e. F.pad(x, (0, 1, 0, 1)) + stride-2 conv.
    g3_splitk (A/B tools): keep the split-K path on the round-2 kernels (gemm3 + splitk_reduce)."""
    _f16(a), _f16(w)
    N, K = w.shape
    c1 = a.shape[1]
    ctot = c1 + (a2.shape[1] if a2 is not None else 0)
    if mode == DENSE:
        M = a.shape[0]
        assert ctot == K, (ctot, K)
        Hs = Ws = Ho = Wo = 0
    else:
        Hs, Ws, Ho, Wo = geom
        M = m_out
        assert K == 9 * ctot, (K, ctot)
    n_out = N // 2 if geglu else N
    if out is None:
        out = empty((M, n_out), a)
    assert out.shape[0] == M and out.shape[1] == n_out
    _untag(out)
    share = gemm_share()
    if tileloop is None:
        tileloop = TILELOOP
    if mode == DENSE and not (tile or deep or cfg or g3_splitk) and (splits is None or splits == 1) and tileloop is not False \
            and (tileloop or _tileloop_wanted(M, N, K, share)):
        if gemm_tileloop(a, w, a2=a2, bias=bias, residual=residual, out=out, alpha=alpha, rows_per_batch=rows_per_batch,
                         geglu=geglu) is not None:
            return (out, None) if gn_hw else out
    flags = tile | (0x200 if geglu else 0) | (0x400 if deep else 0) | (cfg << 12) \
        | (0 if pad_front else 0x800) | (nsplit << 16) | (share << 20) | (0x1000000 if g3_splitk else 0)
    if bias is not None:
        _f32(bias)
        assert bias.shape[-1] == N
    if splits is None:
        if geglu or tile or deep or cfg:
            splits = 1
        else:
            key = (M, N, K, mode, share, lib.is_emulated())
            splits = _SPLIT_PLAN.get(key)
            if splits is None:
                splits = _SPLIT_PLAN[key] = lib.load().mc_gemm_splitk_plan(M, N, K, mode | (share << 8))
            if splits > 1:      # plan = K ranges | (gemm3 geometry << 8)
                flags |= (splits >> 8) << 12
                splits &= 0xFF
    if splits > 1:
        ws = workspace("gemm_splitk", a, M, N, splits)
        lib.call("mc_gemm_splitk_f16", _p(a), _p(a2), _p(w), _p(out), _p(residual), _p(bias), M, N, K, _ld(a),
                 _ld(a2), _ld(out), _ld(residual), c1, ctot, mode, Hs, Ws, Ho, Wo, rows_per_batch, float(alpha),
                 flags, _p(ws), splits, _stream(a))
        return (out, None) if gn_hw else out
    if gn_hw and GN_FROM_EPILOGUE and not (tile or deep or geglu) and cfg in (0, 11, 15) and M % gn_hw == 0 and gn_hw % 32 == 0 and N in (320, 640, 1280):
        part = torch.empty(lib.workspace_bytes("gemm_gnstats", M // gn_hw, gn_hw) // 4, dtype=torch.float32, device=a.device)
        rows = lib.call_count("mc_gemm_gnstats_f16", _p(a), _p(a2), _p(w), _p(out), _p(residual), _p(bias), M, N, K, _ld(a),
                              _ld(a2), _ld(out), _ld(residual), c1, ctot, mode, Hs, Ws, Ho, Wo, rows_per_batch, float(alpha),
                              flags, _p(part), gn_hw, _stream(a))
        if rows:
            return out, (part, gn_hw // rows)
    lib.call("mc_gemm_f16", _p(a), _p(a2), _p(w), _p(out), _p(residual), _p(bias), M, N, K, _ld(a), _ld(a2),
             _ld(out), _ld(residual), c1, ctot, mode, Hs, Ws, Ho, Wo, rows_per_batch, float(alpha), flags,
             _stream(a))
    return (out, None) if gn_hw else out


def _untag(out):
    """a tensor that is written again no longer holds what its GroupNorm statistics were taken from"""
    if out is not None and hasattr(out, "_mc_gnp"):
        del out._mc_gnp


def gnp_of(x, hw):
    """the GroupNorm partial sums the producing GEMM's epilogue left for tensor `x` (attached by `gemm(..., gn_hw=hw)`'s caller
    with `tag_gnp`), or None.  Views / slices / concatenations are new tensor objects and carry none."""
    t = getattr(x, "_mc_gnp", None)
    return t[0] if t is not None and t[1] == hw else None


def tag_gnp(x, gnp, hw):
    if gnp is not None:
        x._mc_gnp = (gnp, hw)
    return x


def pack_conv_k(w_taps):
    """[N, 9, C] (tap-major) -> [N, 9*C] in the kernels' K order: 64-channel tile major, tap minor."""
    N, T, C = w_taps.shape
    assert T == 9 and C % 64 == 0
    return w_taps.reshape(N, 9, C // 64, 64).permute(0, 2, 1, 3).reshape(N, 9 * C).contiguous()


def interleave_geglu(t):
    """rows [h_0..h_{D-1}, g_0..g_{D-1}] -> [h_0, g_0, h_1, g_1, ...] (weight [2D, K] or bias [2D])"""
    D = t.shape[0] // 2
    return torch.stack([t[:D], t[D:]], dim=1).reshape(t.shape).contiguous()


# ---- GroupNorm -------------------------------------------------------------------------------------
def gn_stats(x, x2, frames, hw, eps):
    c1 = x.shape[1]
    ctot = c1 + (x2.shape[1] if x2 is not None else 0)
    partial = workspace("groupnorm", x, frames, hw)
    stats = empty((frames, 32, 2), x, torch.float32)
    lib.call("mc_groupnorm_stats_f16", _p(x), _p(x2), _ld(x), _ld(x2), c1, ctot, frames, hw, float(eps),
             _p(partial), _p(stats), _stream(x))
    return stats


def gn_apply(x, x2, stats, gamma, beta, silu, frames, hw, out=None):
    c1 = x.shape[1]
    ctot = c1 + (x2.shape[1] if x2 is not None else 0)
    if out is None:
        out = empty((frames * hw, ctot), x)
    lib.call("mc_groupnorm_apply_f16", _p(x), _p(x2), _ld(x), _ld(x2), c1, ctot, frames, hw, _p(stats),
             _p(_f32(gamma)), _p(_f32(beta)), _p(out), _ld(out), int(silu), _stream(x))
    return out


def gn_fwd(x, x2, gamma, beta, silu, frames, hw, eps, out=None, gnp=None):
    """GroupNorm(32) (+ SiLU) of [frames * hw, c1 (+ c2)] tokens -> (out, stats); stats = (mean, rstd) per (frame, group) for
    the backward.  gn_stats + gn_apply without the finalize launch (mc_groupnorm_fwd_f16).
    gnp = (partial, chunks per frame) from `gemm(..., gn_hw=hw)` that produced x: ONE launch, no statistics pass."""
    c1 = x.shape[1]
    ctot = c1 + (x2.shape[1] if x2 is not None else 0)
    stats = empty((frames, 32, 2), x, torch.float32)
    if out is None:
        out = empty((frames * hw, ctot), x)
    if gnp is not None and x2 is None:
        lib.call("mc_groupnorm_fwd_partial_f16", _p(x), _ld(x), ctot, frames, hw, float(eps), _p(gnp[0]), gnp[1], _p(stats),
                 _p(_f32(gamma)), _p(_f32(beta)), _p(out), _ld(out), int(silu), _stream(x))
        return out, stats
    partial = workspace("groupnorm", x, frames, hw)
    lib.call("mc_groupnorm_fwd_f16", _p(x), _p(x2), _ld(x), _ld(x2), c1, ctot, frames, hw, float(eps), _p(partial),
             _p(stats), _p(_f32(gamma)), _p(_f32(beta)), _p(out), _ld(out), int(silu), _stream(x))
    return out, stats


def gn_bwd(x, x2, dz, stats, gamma, beta, silu, frames, hw, out=None, accumulate=False):
    c1 = x.shape[1]
    ctot = c1 + (x2.shape[1] if x2 is not None else 0)
    partial = workspace("groupnorm", x, frames, hw)
    bstats = workspace("groupnorm_bwd_stats", x, frames)
    if out is None:
        assert not accumulate
        out = empty((frames * hw, ctot), x)
    lib.call("mc_groupnorm_bwd_f16", _p(x), _p(x2), _ld(x), _ld(x2), c1, ctot, frames, hw, _p(dz), _ld(dz),
             _p(stats), _p(_f32(gamma)), _p(_f32(beta)), int(silu), _p(partial), _p(bstats), _p(out), _ld(out),
             int(accumulate), _stream(x))
    return out


# ---- LayerNorm -------------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, eps=1e-5, pe=None, hw=0, save_stats=True, out=None):
    M, C = x.shape
    if out is None:
        out = empty((M, C), x)
    stats = empty((M, 2), x, torch.float32) if save_stats else None
    nfr = pe.shape[0] if pe is not None else 0
    lib.call("mc_layernorm_fwd_f16", _p(x), _ld(x), _p(out), _ld(out), _p(_f32(gamma)), _p(_f32(beta)),
             _p(_f32(pe)), hw, nfr, _p(stats), M, C, float(eps), _stream(x))
    return out, stats


def layernorm_bwd(dy, x, stats, gamma, add=None, out=None):
    M, C = x.shape
    if out is None:
        out = empty((M, C), x)
    lib.call("mc_layernorm_bwd_f16", _p(dy), _ld(dy), _p(x), _ld(x), _p(stats), _p(_f32(gamma)), _p(add),
             _ld(add), _p(out), _ld(out), M, C, _stream(x))
    return out


# ---- norm + Linear in one launch (K = 320 level) -------------------------------------------------------
# Below this the streaming kernel does not fill the chip.  NOT gemm_api.hip's rule for a plain K = 320 GEMM (that one keeps
# N = 320 on the tiled kernels until M >= 98304): with the norm inside, the fused launch was measured faster than norm + GEMM
# at every (M, N) the engine produces from 32768 rows up - N = 320: 38.1 vs 56.5 us (LayerNorm + q) and 50.0 vs 72.8 us
# (GroupNorm + proj_in) at M = 65536, 25.2 vs 33.3 / 34.9 vs 47.9 at M = 32768 (profiles/r04_norm_gemm_microbench.jsonl);
# M = 65536 is what the shared prefix of the CFG batch runs at config 2 (engine.forward: dup), M = 32768 config 1's B = 2.
NORM_GEMM_MIN_ROWS = 32768
# GroupNorm statistics from the producing GEMM's epilogue (mc_gemm_gnstats_f16); False = always the statistics pass (A/B)
GN_FROM_EPILOGUE = True


def norm_gemm(x, w, kind, gamma, beta, *, bias=None, pe=None, hw=0, eps=1e-5, save_stats=True, geglu=False, out=None,
              stats=None, force=False, gnp=None):
    """LayerNorm (kind 1, + temporal position table `pe` [F, C]) or GroupNorm(32) without activation (kind 2, frames of `hw`
    tokens) of the rows of x [M, 320], followed by the Linear w [N, 320] (+ bias [1, N]; geglu: fused GEGLU epilogue) - ONE
    launch, the normalised rows never reach HBM (mc_norm_gemm_f16).  -> (out, stats), stats as layernorm_fwd / gn_fwd return
    them (None for kind 1 without save_stats); or None when the shape is outside that kernel: the caller then runs the norm
    and the GEMM separately.  `stats` (kind 1): write the row statistics into this [M, 2] fp32 view.  `force`: skip the
    row-count rule (tests)."""
    _f16(x), _f16(w)
    M, K = x.shape
    N = w.shape[0]
    if K != 320 or w.shape[1] != K or N % 32 or (geglu and N % 64) or (M < NORM_GEMM_MIN_ROWS and not force):
        return None
    if kind == 2:
        if hw <= 0 or M % hw or hw % 256:
            return None
        frames = M // hw
        stats = empty((frames, 32, 2), x, torch.float32)
        # gnp (round 6): the sums left by the epilogue of the GEMM that produced x, (partial, chunks per frame): no statistics pass
        partial = gnp[0] if gnp is not None and gnp[1] <= 255 else workspace("groupnorm", x, frames, hw)
    else:
        if pe is not None and (hw <= 0 or hw % 256 or M % hw):
            return None
        if stats is not None:
            assert stats.shape == (M, 2) and stats.dtype == torch.float32 and stats.is_contiguous()
        elif save_stats:
            stats = empty((M, 2), x, torch.float32)
        partial = None
    n_out = N // 2 if geglu else N
    if out is None:
        out = empty((M, n_out), x)
    if bias is not None:
        _f32(bias)
        assert bias.shape[-1] == N and bias.numel() == N
    ok = lib.try_call("mc_norm_gemm_f16", _p(x), _p(w), _p(out), _p(bias), M, N, K, _ld(x), _ld(out), kind,
                      _p(_f32(gamma)), _p(_f32(beta)), _p(_f32(pe)) if kind == 1 else None, hw,
                      pe.shape[0] if (kind == 1 and pe is not None) else 0, float(eps), _p(stats), _p(partial),
                      (0x200 if geglu else 0) | ((gnp[1] << 16) if (kind == 2 and gnp is not None and gnp[1] <= 255) else 0),
                      _stream(x))
    return (out, stats) if ok else None


# ---- spatial / cross attention ----------------------------------------------------------------------
def attn_fwd(q, k, v, Nq, Nk, heads, d, nbatch, kv_bdiv=1, scale=None, need_lse=True, out=None):
    scale = d ** -0.5 if scale is None else scale
    if out is None:
        out = empty((q.shape[0], heads * d), q)
    lse = empty((nbatch, heads, Nq), q, torch.float32) if need_lse else None
    lib.call("mc_attn_fwd_f16", _p(q), _p(k), _p(v), _ld(q), _ld(k), _ld(v), _p(out), _ld(out), _p(lse), Nq, Nk,
             heads, d, nbatch, kv_bdiv, float(scale), _stream(q))
    return out, lse


def attn_fwd_causal(q, k, v, N, heads, d, nbatch, scale=None, out=None):
    """causal self-attention (CLIP text encoder): tokens [(nbatch N), heads*d]"""
    scale = d ** -0.5 if scale is None else scale
    if out is None:
        out = empty((q.shape[0], heads * d), q)
    lib.call("mc_attn_fwd_causal_f16", _p(q), _p(k), _p(v), _ld(q), _ld(k), _ld(v), _p(out), _ld(out), N, heads, d,
             nbatch, float(scale), _stream(q))
    return out


def clip_embed(ids, tok, pos):
    """ids int64 [B, S] -> fp16 [(B S), C] = token_embedding[ids] + position_embedding[:S]"""
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    B, S = ids.shape
    out = empty((B * S, tok.shape[1]), tok)
    lib.call("mc_clip_embed_f16", _p(ids), _p(_f16(tok)), _p(_f16(pos)), _p(out), B, S, tok.shape[1], tok.shape[0], _stream(tok))
    return out


def quick_gelu(x):
    out = torch.empty_like(x)
    lib.call("mc_quick_gelu_f16", _p(_f16(x)), _p(out), x.numel(), _stream(x))
    return out


def attn_bwd(q, k, v, o, do, lse, Nq, Nk, heads, d, nbatch, kv_bdiv=1, scale=None, dq=None, dk=None, dv=None,
             need_dkv=True):
    scale = d ** -0.5 if scale is None else scale
    dbuf = workspace("attn_bwd", q, nbatch, heads, Nq)
    if dq is None:
        dq = empty((q.shape[0], heads * d), q)
    if need_dkv:
        if dk is None:
            dk = empty((k.shape[0], heads * d), q)
        if dv is None:
            dv = empty((v.shape[0], heads * d), q)
    lib.call("mc_attn_bwd_f16", _p(q), _p(k), _p(v), _ld(q), _ld(k), _ld(v), _p(o), _ld(o), _p(do), _ld(do),
             _p(lse), _p(dbuf), _p(dq), _ld(dq), _p(dk) if need_dkv else None, _ld(dk) if need_dkv else 0,
             _p(dv) if need_dkv else None, _ld(dv) if need_dkv else 0, Nq, Nk, heads, d, nbatch, kv_bdiv,
             float(scale), _stream(q))
    return dq, dk, dv


# ---- temporal attention + guidance --------------------------------------------------------------------
def tattn_fwd(q, k, v, B, F, HW, heads, d, scale=None, out=None):
    scale = d ** -0.5 if scale is None else scale
    if out is None:
        out = empty((q.shape[0], heads * d), q)
    assert _ld(q) == _ld(k) == _ld(v)
    lib.call("mc_tattn_fwd_f16", _p(q), _p(k), _p(v), _ld(q), _p(out), _ld(out), B, F, HW, heads, d,
             float(scale), _stream(q))
    return out


def tattn_top1(q, k, B, F, HW, heads, d, scale=None):
    """-> (values fp16 [B*HW, heads, F, 1], indices uint8 [B*HW, heads, F, 1]) as stored by the reference."""
    scale = d ** -0.5 if scale is None else scale
    assert _ld(q) == _ld(k)
    val = empty((B * HW, heads, F, 1), q)
    idx = empty((B * HW, heads, F, 1), q, torch.uint8)
    lib.call("mc_tattn_top1_f16", _p(q), _p(k), _ld(q), _p(val), _p(idx), B, F, HW, heads, d, float(scale),
             _stream(q))
    return val, idx


def tattn_prob(q, k, B, F, HW, heads, d, scale=None):
    """-> P fp16 [B*HW, heads, F, F] (get_temp_attn_prob, motionclone_functions.py:260-283)"""
    scale = d ** -0.5 if scale is None else scale
    prob = empty((B * HW, heads, F, F), q)
    lib.call("mc_tattn_prob_f16", _p(q), _p(k), _ld(q), _p(prob), B, F, HW, heads, d, float(scale), _stream(q))
    return prob


def tattn_loss(q, k, ref_idx, ref_val, B, F, HW, heads, d, scale=None):
    scale = d ** -0.5 if scale is None else scale
    assert ref_idx.dtype == torch.uint8 and ref_idx.is_contiguous()
    _f32(ref_val)
    ul = workspace("tattn_loss", q, B, HW, heads)
    loss = empty((1,), q, torch.float32)
    lib.call("mc_tattn_loss_f16", _p(q), _p(k), _ld(q), _p(ref_idx), _p(ref_val), _p(ul), _p(loss), B, F, HW,
             heads, d, float(scale), _stream(q))
    return loss


def tattn_bwd(q, k, v, do, dq, dk, dv, B, F, HW, heads, d, ref_idx=None, ref_val=None, seed_coef=0.0,
              scale=None):
    scale = d ** -0.5 if scale is None else scale
    assert _ld(q) == _ld(k) == _ld(v) and _ld(dq) == _ld(dk) == _ld(dv)
    if ref_idx is not None:
        assert ref_idx.dtype == torch.uint8 and ref_idx.is_contiguous()
        _f32(ref_val)
    lib.call("mc_tattn_bwd_f16", _p(q), _p(k), _p(v), _ld(q), _p(do), _ld(do), _p(dq), _p(dk), _p(dv), _ld(dq),
             _p(ref_idx), _p(ref_val), float(seed_coef), B, F, HW, heads, d, float(scale), _stream(q))


# ---- element-wise ------------------------------------------------------------------------------------
def geglu_fwd(x, out=None):
    M, D2 = x.shape
    D = D2 // 2
    if out is None:
        out = empty((M, D), x)
    lib.call("mc_geglu_fwd_f16", _p(x), _ld(x), _p(out), _ld(out), M, D, _stream(x))
    return out


def geglu_bwd(dout, x, out=None):
    M, D2 = x.shape
    if out is None:
        out = empty((M, D2), x)
    lib.call("mc_geglu_bwd_f16", _p(dout), _ld(dout), _p(x), _ld(x), _p(out), _ld(out), M, D2 // 2, _stream(x))
    return out


def add(a, b=None, out=None, sa=1.0, sb=1.0):
    M, C = a.shape
    if out is None:
        out = empty((M, C), a)
    _untag(out)
    lib.call("mc_add_f16", _p(a), _ld(a), _p(b), _ld(b), _p(out), _ld(out), M, C, float(sa), float(sb), _stream(a))
    return out


def dup_rows(x):
    """[x | x] along the rows (the shared prefix of the CFG batch hands the same rows to both halves, engine._spatial): two
    launches of the library's own copy (mc_add_f16 with one operand) - no framework kernel inside a step"""
    M, C = x.shape
    out = empty((2 * M, C), x)
    add(x, None, out=out[:M])
    add(x, None, out=out[M:])
    return out


def sumpool2(x, frames, H, W, out=None, accumulate=False):
    C = x.shape[1]
    if out is None:
        out = empty((frames * H * W, C), x)
    lib.call("mc_sumpool2_f16", _p(x), _ld(x), _p(out), _ld(out), frames, H, W, C, int(accumulate), _stream(x))
    return out


def latent_to_cl(lat, cp=64):
    B, CL, F, H, W = lat.shape
    lat = lat.contiguous()
    out = empty((B * F * H * W, cp), lat)
    lib.call("mc_latent_to_cl_f16", _p(lat), _p(out), B, CL, F, H * W, cp, _stream(lat))
    return out


def cl_to_latent(x, B, CL, F, H, W, scale=1.0, f32=False):
    out = empty((B, CL, F, H, W), x, torch.float32 if f32 else torch.float16)
    lib.call("mc_cl_to_latent_f16", _p(x), _ld(x), _p(out), int(f32), float(scale), B, CL, F, H * W, _stream(x))
    return out


def softmax_rows_(x):
    """in-place softmax over the last dim of an fp16 [rows, cols] matrix (fp32 arithmetic)"""
    _f16(x)
    lib.call("mc_softmax_rows_f16", _p(x), _ld(x), x.shape[0], x.shape[1], _stream(x))
    return x


def video_post(tokens, C, F, H, W):
    """channels-last decoded frames [(f y x), ld] -> float32 [1, C, F, H, W], (x / 2 + 0.5).clamp(0, 1)"""
    out = empty((1, C, F, H, W), tokens, torch.float32)
    lib.call("mc_video_post_f32", _p(tokens), _ld(tokens), _p(out), C, F, H * W, _stream(tokens))
    return out


def video_resize(frames_u8, H, W, quantise=2):
    """uint8 [N, Hs, Ws, 3] (device) -> fp16 [N, 3, H, W] in [-1, 1]: bilinear align_corners=True, / 127.5 - 1.
    quantise: 2 (default) = BIT-EXACT with torch's CPU uint8 resize (what util.py:232-238 runs on the decord frames: separable
    fixed-point passes, checked against the installed torch); 0 = no re-quantisation (float result), 1 = float result rounded
    to nearest level, 3 = float result truncated (the generic-template semantics of torch 2.0.x as far as its source tells -
    not verifiable offline)."""
    assert frames_u8.dtype == torch.uint8 and frames_u8.dim() == 4 and frames_u8.shape[3] == 3
    frames_u8 = frames_u8.contiguous()
    N, Hs, Ws, _ = frames_u8.shape
    out = torch.empty((N, 3, H, W), dtype=torch.float16, device=frames_u8.device)
    lib.call("mc_video_resize_u8_f16", _p(frames_u8), _p(out), N, Hs, Ws, H, W, int(2 if quantise is True else quantise), _stream(frames_u8))
    return out


def vae_sample(moments, noise, lat):
    """moment tokens [(n h w), 2*lat] + N(0,1) draw [n, lat, h, w] -> mean + std * noise  [n, lat, h, w]"""
    _f16(moments)
    n, _, h, w = noise.shape
    noise = _f16(noise).contiguous()
    out = torch.empty_like(noise)
    lib.call("mc_vae_sample_f16", _p(moments), _ld(moments), _p(noise), _p(out), n, lat, h * w, _stream(moments))
    return out


def vae_mode(moments, n, lat, h, w):
    _f16(moments)
    out = empty((n, lat, h, w), moments)
    lib.call("mc_vae_sample_f16", _p(moments), _ld(moments), None, _p(out), n, lat, h * w, _stream(moments))
    return out


def timestep_embed(t, dim, like):
    out = empty((t.shape[0], dim), like)
    lib.call("mc_timestep_embed_f16", _p(_f32(t)), _p(out), t.shape[0], dim, _stream(like))
    return out


def silu(x):
    out = torch.empty_like(x)
    lib.call("mc_silu_f16", _p(x), _p(out), x.numel(), _stream(x))
    return out


def ddim_step_general(sample, model_output, score, noise, coef, want_prev=True, want_x0=True, want_eps=False):
    """mc_ddim_step_general_f16: every branch of schedule_customized_step as one elementwise pass.  All tensors share one
    layout ([B, C, F, H, W] as the reference holds them); `coef` = dict(x0_s, x0_m, ep_s, ep_m, clip, rederive, sqrt_a,
    sqrt_b, score_coef, c_x0, c_dir, c_noise).  Returns (prev, x0, eps) with None for outputs not asked for."""
    sample = sample.contiguous().half()
    model_output = model_output.contiguous().half()
    if tuple(model_output.shape) != tuple(sample.shape):
        raise ValueError("ddim_step_general: sample %s vs model_output %s" % (tuple(sample.shape), tuple(model_output.shape)))
    for name, t in (("score", score), ("variance_noise", noise)):
        if t is not None and tuple(t.shape) != tuple(sample.shape):
            raise ValueError("ddim_step_general: %s %s vs sample %s" % (name, tuple(t.shape), tuple(sample.shape)))
    score = None if score is None else score.float().contiguous()
    noise = None if noise is None else noise.contiguous().half()
    prev = torch.empty_like(sample) if want_prev else None
    x0 = torch.empty_like(sample) if want_x0 else None
    eps = torch.empty_like(sample) if want_eps else None
    c = coef
    lib.call("mc_ddim_step_general_f16", _p(sample), _p(model_output), _p(score), _p(noise), _p(prev), _p(x0), _p(eps),
             sample.numel(), float(c["x0_s"]), float(c["x0_m"]), float(c["ep_s"]), float(c["ep_m"]), float(c.get("clip", 0.0)),
             int(bool(c.get("rederive", False))), float(c["sqrt_a"]), float(c["sqrt_b"]), float(c.get("score_coef", 0.0)),
             float(c["c_x0"]), float(c["c_dir"]), float(c.get("c_noise", 0.0)), _stream(sample))
    return prev, x0, eps


def cfg_ddim_step(eps_c, eps_u, x, score, cfg, a_t, a_prev, score_coef, want_eps=False, sigma=0.0):
    """x, score: [1, CL, F, H, W]; eps_*: channels-last token matrices (first CL columns).  sigma = eta * sqrt(variance)
    of schedule_customized_step (:364-365): the direction coefficient becomes sqrt(1 - a_prev - sigma^2) (:386); the
    caller adds sigma * noise (:391-405)."""
    B, CL, F, H, W = x.shape
    if B != 1 or (score is not None and tuple(score.shape) != tuple(x.shape)):
        raise ValueError("cfg_ddim_step updates one video per call: x %s, score %s" % (tuple(x.shape), None if score is None else tuple(score.shape)))
    x = x.contiguous()
    out = torch.empty_like(x)
    eps_out = torch.empty_like(x) if want_eps else None
    if score is not None:
        score = _f32(score.contiguous())
    assert _ld(eps_c) == _ld(eps_u)
    lib.call("mc_cfg_ddim_step_f16", _p(eps_c), _p(eps_u), _ld(eps_c), _p(x), _p(score), _p(out), _p(eps_out),
             float(cfg), float(a_t) ** 0.5, float(1.0 - a_t) ** 0.5, float(a_prev) ** 0.5,
             max(float(1.0 - a_prev) - float(sigma) ** 2, 0.0) ** 0.5, float(score_coef), CL, F, H * W, _stream(x))
    return (out, eps_out) if want_eps else out
