"""ORACLE - TEST INFRASTRUCTURE ONLY.  Never imported by motionclone_amd / motionclone.

CPU (plain PyTorch, fp32) restatement of the reference's UNet3D forward for the guided-DDIM hot
path, written functionally over a flat state-dict that uses the reference's parameter names, so
that the very same weights can be fed to (a) the reference's own modules, (b) this restatement and
(c) the HIP engine.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it.

Parity pin: tests/test_oracle_pins.py runs this file against the *unmodified* reference code
(imported from /root/reference through oracle/reference_shim.py) on seeded random weights, and
tests/golden/*.pt hold outputs of the reference itself (made by tests/golden/make_golden.py) that
travel to the GPU box.  The reference ships no tests or golden vectors of its own.

Every function cites the reference lines it restates (paths relative to the reference root).
Tensors use the reference's layout [B, C, F, H, W].
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as Fn

SD15_CONFIG = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    cross_attention_dim=768, attention_heads=8, norm_num_groups=32, norm_eps=1e-5,
    down_has_attn=(True, True, True, False), up_has_attn=(False, True, True, True),
    motion_heads=8, motion_pe_max_len=32, motion_mid_block=False,
)

TINY_CONFIG = dict(
    in_channels=4, out_channels=4, block_out_channels=(64, 128, 128, 128), layers_per_block=2,
    cross_attention_dim=64, attention_heads=2, norm_num_groups=32, norm_eps=1e-5,
    down_has_attn=(True, True, True, False), up_has_attn=(False, True, True, True),
    motion_heads=2, motion_pe_max_len=32, motion_mid_block=False,
)


# ---- parameter inventory (names = reference state_dict keys) --------------------------------------
def _resnet_shapes(p, cin, cout, temb):
    s = OrderedDict()
    s[p + "norm1.weight"] = (cin,)
    s[p + "norm1.bias"] = (cin,)
    s[p + "conv1.weight"] = (cout, cin, 3, 3)
    s[p + "conv1.bias"] = (cout,)
    s[p + "time_emb_proj.weight"] = (cout, temb)
    s[p + "time_emb_proj.bias"] = (cout,)
    s[p + "norm2.weight"] = (cout,)
    s[p + "norm2.bias"] = (cout,)
    s[p + "conv2.weight"] = (cout, cout, 3, 3)
    s[p + "conv2.bias"] = (cout,)
    if cin != cout:
        s[p + "conv_shortcut.weight"] = (cout, cin, 1, 1)
        s[p + "conv_shortcut.bias"] = (cout,)
    return s


def _attn_shapes(p, c, kv_dim):
    s = OrderedDict()
    s[p + "to_q.weight"] = (c, c)
    s[p + "to_k.weight"] = (c, kv_dim)
    s[p + "to_v.weight"] = (c, kv_dim)
    s[p + "to_out.0.weight"] = (c, c)
    s[p + "to_out.0.bias"] = (c,)
    return s


def _ff_shapes(p, c):
    s = OrderedDict()
    s[p + "net.0.proj.weight"] = (8 * c, c)
    s[p + "net.0.proj.bias"] = (8 * c,)
    s[p + "net.2.weight"] = (c, 4 * c)
    s[p + "net.2.bias"] = (c,)
    return s


def _spatial_shapes(p, c, xdim):
    s = OrderedDict()
    s[p + "norm.weight"] = (c,)
    s[p + "norm.bias"] = (c,)
    s[p + "proj_in.weight"] = (c, c, 1, 1)
    s[p + "proj_in.bias"] = (c,)
    b = p + "transformer_blocks.0."
    s.update(_attn_shapes(b + "attn1.", c, c))
    s[b + "norm1.weight"] = (c,)
    s[b + "norm1.bias"] = (c,)
    s.update(_attn_shapes(b + "attn2.", c, xdim))
    s[b + "norm2.weight"] = (c,)
    s[b + "norm2.bias"] = (c,)
    s.update(_ff_shapes(b + "ff.", c))
    s[b + "norm3.weight"] = (c,)
    s[b + "norm3.bias"] = (c,)
    s[p + "proj_out.weight"] = (c, c, 1, 1)
    s[p + "proj_out.bias"] = (c,)
    return s


def _motion_shapes(p, c):
    p = p + "temporal_transformer."
    s = OrderedDict()
    s[p + "norm.weight"] = (c,)
    s[p + "norm.bias"] = (c,)
    s[p + "proj_in.weight"] = (c, c)
    s[p + "proj_in.bias"] = (c,)
    b = p + "transformer_blocks.0."
    for a in range(2):
        s.update(_attn_shapes(b + "attention_blocks.%d." % a, c, c))
    for a in range(2):
        s[b + "norms.%d.weight" % a] = (c,)
        s[b + "norms.%d.bias" % a] = (c,)
    s.update(_ff_shapes(b + "ff.", c))
    s[b + "ff_norm.weight"] = (c,)
    s[b + "ff_norm.bias"] = (c,)
    s[p + "proj_out.weight"] = (c, c)
    s[p + "proj_out.bias"] = (c,)
    return s


def param_shapes(cfg):
    """All parameters of the reference UNet3DConditionModel (unet.py:38-249) for `cfg`."""
    ch = cfg["block_out_channels"]
    temb = ch[0] * 4
    xdim = cfg["cross_attention_dim"]
    L = cfg["layers_per_block"]
    s = OrderedDict()
    s["conv_in.weight"] = (ch[0], cfg["in_channels"], 3, 3)
    s["conv_in.bias"] = (ch[0],)
    s["time_embedding.linear_1.weight"] = (temb, ch[0])
    s["time_embedding.linear_1.bias"] = (temb,)
    s["time_embedding.linear_2.weight"] = (temb, temb)
    s["time_embedding.linear_2.bias"] = (temb,)
    out = ch[0]
    for i in range(4):
        cin, out = out, ch[i]
        for j in range(L):
            s.update(_resnet_shapes("down_blocks.%d.resnets.%d." % (i, j), cin if j == 0 else out, out, temb))
            if cfg["down_has_attn"][i]:
                s.update(_spatial_shapes("down_blocks.%d.attentions.%d." % (i, j), out, xdim))
            s.update(_motion_shapes("down_blocks.%d.motion_modules.%d." % (i, j), out))
        if i < 3:
            s["down_blocks.%d.downsamplers.0.conv.weight" % i] = (out, out, 3, 3)
            s["down_blocks.%d.downsamplers.0.conv.bias" % i] = (out,)
    c = ch[-1]
    s.update(_resnet_shapes("mid_block.resnets.0.", c, c, temb))
    s.update(_spatial_shapes("mid_block.attentions.0.", c, xdim))
    s.update(_resnet_shapes("mid_block.resnets.1.", c, c, temb))
    rev = list(reversed(ch))
    out = rev[0]
    for i in range(4):
        prev, out = out, rev[i]
        cin = rev[min(i + 1, 3)]
        for j in range(L + 1):
            skip = cin if j == L else out
            rin = prev if j == 0 else out
            s.update(_resnet_shapes("up_blocks.%d.resnets.%d." % (i, j), rin + skip, out, temb))
            if cfg["up_has_attn"][i]:
                s.update(_spatial_shapes("up_blocks.%d.attentions.%d." % (i, j), out, xdim))
            s.update(_motion_shapes("up_blocks.%d.motion_modules.%d." % (i, j), out))
        if i < 3:
            s["up_blocks.%d.upsamplers.0.conv.weight" % i] = (out, out, 3, 3)
            s["up_blocks.%d.upsamplers.0.conv.bias" % i] = (out,)
    s["conv_norm_out.weight"] = (ch[0],)
    s["conv_norm_out.bias"] = (ch[0],)
    s["conv_out.weight"] = (cfg["out_channels"], ch[0], 3, 3)
    s["conv_out.bias"] = (cfg["out_channels"],)
    return s


def random_state_dict(cfg, seed=1234, dtype=torch.float32):
    """Seeded synthetic weights (no checkpoints exist offline): PyTorch-default-like fan-in uniform for
    conv/linear, unit norms; motion proj_out ~ N(0, 0.02) instead of the zero init of
    motion_module.py:77-78 so that the temporal path contributes."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, shape in param_shapes(cfg).items():
        if "norm" in name and len(shape) == 1:
            t = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
            t = t + 0.05 * torch.randn(shape, generator=g)
        elif "temporal_transformer.proj_out" in name:
            t = 0.02 * torch.randn(shape, generator=g)
        else:
            if name.endswith("bias"):
                wshape = param_shapes_cache(cfg)[name[:-4] + "weight"]
            else:
                wshape = shape
            fan_in = 1
            for d in wshape[1:]:
                fan_in *= d
            bound = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        sd[name] = t.to(dtype)
    return sd


_SHAPE_CACHE = {}


def param_shapes_cache(cfg):
    key = repr(sorted(cfg.items()))
    if key not in _SHAPE_CACHE:
        _SHAPE_CACHE[key] = param_shapes(cfg)
    return _SHAPE_CACHE[key]


# ---- building blocks --------------------------------------------------------------------------------
def _to_frames(x):
    B, C, F, H, W = x.shape
    return x.permute(0, 2, 1, 3, 4).reshape(B * F, C, H, W)


def _from_frames(x, B):
    BF, C, H, W = x.shape
    return x.reshape(B, BF // B, C, H, W).permute(0, 2, 1, 3, 4)


def _conv(sd, p, x, stride=1, pad=1):
    """InflatedConv3d: per-frame Conv2d (resnet.py:10-18)."""
    B = x.shape[0]
    return _from_frames(Fn.conv2d(_to_frames(x), sd[p + "weight"], sd[p + "bias"], stride=stride, padding=pad), B)


def _gn(sd, p, x, groups, eps):
    """InflatedGroupNorm: per-frame statistics (resnet.py:21-29)."""
    B = x.shape[0]
    return _from_frames(Fn.group_norm(_to_frames(x), groups, sd[p + "weight"], sd[p + "bias"], eps), B)


def _lin(sd, p, x):
    return Fn.linear(x, sd[p + "weight"], sd.get(p + "bias"))


def _ln(sd, p, x):
    return Fn.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], 1e-5)


def resnet_block(sd, p, x, temb, cfg):
    """ResnetBlock3D.forward (resnet.py:183-213; identical closure conv_layer.py:3-50)."""
    G, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    h = _conv(sd, p + "conv1.", Fn.silu(_gn(sd, p + "norm1.", x, G, eps)))
    h = h + _lin(sd, p + "time_emb_proj.", Fn.silu(temb))[:, :, None, None, None]
    h = _conv(sd, p + "conv2.", Fn.silu(_gn(sd, p + "norm2.", h, G, eps)))
    if p + "conv_shortcut.weight" in sd:
        x = _conv(sd, p + "conv_shortcut.", x, pad=0)
    return x + h


MHA_MAX_SCORE_BYTES = 12 << 30   # attention is evaluated in batch chunks above this score-matrix size


class _ChunkedAttention(torch.autograd.Function):
    """softmax(q k^T scale) v per (batch, head) slice, a few batch entries (= frames) at a time, WITH a backward that
    recomputes the probabilities chunk by chunk instead of keeping them: the reference's `_attention` materialises
    [B*heads, N, N] at once and autograd keeps it (attention.py:461-490) - 87 GB in fp32 per level-0 self-attention at
    32 f x 96 x 96, two of them inside the differentiated half.  Every slice's arithmetic is what the unchunked path
    computes (same matmuls, same softmax, torch's own softmax-backward formula); tests/test_oracle_pins.py holds the two
    paths against each other, forward and gradients."""

    @staticmethod
    def forward(ctx, qh, kh, vh, scale, step):
        ctx.save_for_backward(qh, kh, vh)
        ctx.scale, ctx.step = scale, step
        out = [torch.softmax(qh[i:i + step] @ kh[i:i + step].transpose(-1, -2) * scale, dim=-1) @ vh[i:i + step]
               for i in range(0, qh.shape[0], step)]
        return torch.cat(out, 0)

    @staticmethod
    def backward(ctx, go):
        qh, kh, vh = ctx.saved_tensors
        scale, step = ctx.scale, ctx.step
        dq, dk, dv = torch.empty_like(qh), torch.empty_like(kh), torch.empty_like(vh)
        for i in range(0, qh.shape[0], step):
            q, k, v, g = qh[i:i + step], kh[i:i + step], vh[i:i + step], go[i:i + step]
            p = torch.softmax(q @ k.transpose(-1, -2) * scale, dim=-1)
            dv[i:i + step] = p.transpose(-1, -2) @ g
            dp = g @ v.transpose(-1, -2)
            ds = (dp - (dp * p).sum(-1, keepdim=True)) * p          # softmax backward (torch: _softmax_backward_data)
            del dp, p
            ds = ds * scale
            dq[i:i + step] = ds @ k
            dk[i:i + step] = ds.transpose(-1, -2) @ q
            del ds
        return dq, dk, dv, None, None


def _mha(q, k, v, heads):
    """CrossAttention._attention with head split/merge (attention.py:367-379,461-490)."""
    Bq, Nq, C = q.shape
    d = C // heads
    qh = q.reshape(Bq, Nq, heads, d).transpose(1, 2)
    kh = k.reshape(Bq, -1, heads, d).transpose(1, 2)
    vh = v.reshape(Bq, -1, heads, d).transpose(1, 2)
    score_bytes = Bq * heads * Nq * kh.shape[2] * q.element_size()
    if score_bytes > MHA_MAX_SCORE_BYTES:
        # same arithmetic per (batch, head) slice, evaluated a few batch elements at a time (and recomputed in the backward)
        step = max(1, int(Bq * MHA_MAX_SCORE_BYTES // score_bytes))
        return _ChunkedAttention.apply(qh, kh, vh, d ** -0.5, step).transpose(1, 2).reshape(Bq, Nq, C)
    p = torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, dim=-1)
    return (p @ vh).transpose(1, 2).reshape(Bq, Nq, C)


def _feed_forward(sd, p, x):
    """diffusers FeedForward(activation_fn='geglu'): Linear(C,8C) -> h*gelu(gate) -> Linear(4C,C)."""
    h, gate = _lin(sd, p + "net.0.proj.", x).chunk(2, dim=-1)
    return _lin(sd, p + "net.2.", h * Fn.gelu(gate))


def spatial_transformer(sd, p, x, text, cfg):
    """Transformer3DModel.forward + BasicTransformerBlock.forward (attention.py:95-142,256-300)."""
    B, C, F, H, W = x.shape
    heads = cfg["attention_heads"]
    res = x
    h = _gn(sd, p + "norm.", x, cfg["norm_num_groups"], 1e-6)
    h = _conv(sd, p + "proj_in.", h, pad=0)
    tok = h.permute(0, 2, 3, 4, 1).reshape(B * F, H * W, C)      # (b f) (h w) c
    ctx = text.repeat_interleave(F, dim=0)                      # attention.py:100
    b = p + "transformer_blocks.0."
    n = _ln(sd, b + "norm1.", tok)
    a = _mha(_lin(sd, b + "attn1.to_q.", n), _lin(sd, b + "attn1.to_k.", n), _lin(sd, b + "attn1.to_v.", n), heads)
    tok = _lin(sd, b + "attn1.to_out.0.", a) + tok
    n = _ln(sd, b + "norm2.", tok)
    a = _mha(_lin(sd, b + "attn2.to_q.", n), _lin(sd, b + "attn2.to_k.", ctx), _lin(sd, b + "attn2.to_v.", ctx), heads)
    tok = _lin(sd, b + "attn2.to_out.0.", a) + tok
    tok = _feed_forward(sd, b + "ff.", _ln(sd, b + "norm3.", tok)) + tok
    h = tok.reshape(B, F, H, W, C).permute(0, 4, 1, 2, 3)
    return _conv(sd, p + "proj_out.", h, pad=0) + res


def temporal_pe(max_len, dim, device):
    """PositionalEncoding table (motion_module.py:237-243)."""
    pos = torch.arange(max_len, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * (-math.log(10000.0) / dim))
    pe = torch.zeros(max_len, dim)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.to(device)


def motion_module(sd, p, x, cfg, record=None, name=None):
    """VanillaTemporalModule -> TemporalTransformer3DModel -> TemporalTransformerBlock ->
    VersatileAttention (motion_module.py:80-85,137-161,213-225,274-345).  `record[name.attention_blocks.a]`
    receives the pre-head-split (query, key) [(b n), F, C] exactly as MySelfAttnProcessor.record_qkv does
    (xformer_attention.py:31-34)."""
    B, C, F, H, W = x.shape
    heads = cfg["motion_heads"]
    p = p + "temporal_transformer."
    res = x
    h = _gn(sd, p + "norm.", x, cfg["norm_num_groups"], 1e-6)
    tok = h.permute(0, 2, 3, 4, 1).reshape(B * F, H * W, C)
    tok = _lin(sd, p + "proj_in.", tok)
    b = p + "transformer_blocks.0."
    pe = temporal_pe(cfg["motion_pe_max_len"], C, x.device).to(x.dtype)
    N = H * W
    for a in range(2):
        n = _ln(sd, b + "norms.%d." % a, tok)
        seq = n.reshape(B, F, N, C).permute(0, 2, 1, 3).reshape(B * N, F, C) + pe[None, :F]   # (b n) f c
        ap = b + "attention_blocks.%d." % a
        q, k, v = _lin(sd, ap + "to_q.", seq), _lin(sd, ap + "to_k.", seq), _lin(sd, ap + "to_v.", seq)
        if record is not None:
            record[name + ".temporal_transformer.transformer_blocks.0.attention_blocks.%d" % a] = (q, k)
        o = _lin(sd, ap + "to_out.0.", _mha(q, k, v, heads))
        tok = o.reshape(B, N, F, C).permute(0, 2, 1, 3).reshape(B * F, N, C) + tok
    tok = _feed_forward(sd, b + "ff.", _ln(sd, b + "ff_norm.", tok)) + tok
    tok = _lin(sd, p + "proj_out.", tok)
    return tok.reshape(B, F, H, W, C).permute(0, 4, 1, 2, 3) + res


def timestep_embedding(sd, t, dim, dtype):
    """Timesteps(dim, flip_sin_to_cos=True, freq_shift=0) + TimestepEmbedding (unet.py:101-104,386-392)."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    e = t.float()[:, None] * freqs[None]
    emb = torch.cat([torch.cos(e), torch.sin(e)], dim=-1).to(dtype)
    return _lin(sd, "time_embedding.linear_2.", Fn.silu(_lin(sd, "time_embedding.linear_1.", emb)))


def unet_forward(sd, cfg, sample, timestep, text, guidance_block=1, only_motion_feature=False, record=None,
                 hooked=("up_blocks.1",), down_residuals=None, mid_residual=None):
    """unet_customized_forward (motionclone_functions.py:478-662).

    Up blocks with index <= guidance_block stay in the autograd graph, later ones run under no_grad
    (:601-652); `only_motion_feature` returns after the guidance block (:627-628).  `record` collects
    (query, key) of the temporal attentions whose name contains one of `hooked` (util.py:434-440)."""
    B = sample.shape[0]
    L = cfg["layers_per_block"]
    t = torch.as_tensor(timestep, device=sample.device).reshape(-1).expand(B)
    temb = timestep_embedding(sd, t, cfg["block_out_channels"][0], sample.dtype)

    def rec_for(name):
        return record if (record is not None and any(hk in name for hk in hooked)) else None

    x = _conv(sd, "conv_in.", sample)
    skips = [x]
    for i in range(4):
        for j in range(L):
            x = resnet_block(sd, "down_blocks.%d.resnets.%d." % (i, j), x, temb, cfg)
            if cfg["down_has_attn"][i]:
                x = spatial_transformer(sd, "down_blocks.%d.attentions.%d." % (i, j), x, text, cfg)
            nm = "down_blocks.%d.motion_modules.%d" % (i, j)
            x = motion_module(sd, nm + ".", x, cfg, rec_for(nm), nm)
            skips.append(x)
        if i < 3:
            x = _conv(sd, "down_blocks.%d.downsamplers.0.conv." % i, x, stride=2)
            skips.append(x)
    if down_residuals is not None:  # :581-587
        skips = [s + (r.unsqueeze(2) if r.dim() == 4 else r) for s, r in zip(skips, down_residuals)]
    x = resnet_block(sd, "mid_block.resnets.0.", x, temb, cfg)
    x = spatial_transformer(sd, "mid_block.attentions.0.", x, text, cfg)
    x = resnet_block(sd, "mid_block.resnets.1.", x, temb, cfg)
    if mid_residual is not None:    # :595-598
        x = x + (mid_residual.unsqueeze(2) if mid_residual.dim() == 4 else mid_residual)

    def up_block(i, x):
        for j in range(L + 1):
            x = torch.cat([x, skips.pop()], dim=1)   # unet_blocks.py:632-634
            x = resnet_block(sd, "up_blocks.%d.resnets.%d." % (i, j), x, temb, cfg)
            if cfg["up_has_attn"][i]:
                x = spatial_transformer(sd, "up_blocks.%d.attentions.%d." % (i, j), x, text, cfg)
            nm = "up_blocks.%d.motion_modules.%d" % (i, j)
            x = motion_module(sd, nm + ".", x, cfg, rec_for(nm), nm)
        if i < 3:
            Bx = x.shape[0]
            u = Fn.interpolate(_to_frames(x), scale_factor=2.0, mode="nearest")   # resnet.py:65
            x = _conv(sd, "up_blocks.%d.upsamplers.0.conv." % i, _from_frames(u, Bx))
        return x

    for i in range(4):
        if i <= guidance_block:
            x = up_block(i, x)
        else:
            if only_motion_feature:
                return None
            with torch.no_grad():
                x = up_block(i, x)
    x = Fn.silu(_gn(sd, "conv_norm_out.", x, cfg["norm_num_groups"], cfg["norm_eps"]))
    return _conv(sd, "conv_out.", x)


# ---- SparseCtrl (image-to-video conditioning encoder) ------------------------------------------------------
COND_EMBEDDING_CHANNELS = (16, 32, 96, 256)   # conditioning_embedding_out_channels default (sparse_controlnet.py:124)


def controlnet_param_shapes(cfg, conditioning_channels=4, simplified=True, embedding_channels=COND_EMBEDDING_CHANNELS):
    """Parameters of the reference SparseControlNetModel (sparse_controlnet.py:150-314): motion modules with a single
    Temporal_Self attention, 12 + 1 zero-initialised 1x1 output convs, and the condition embedding of either
    configs/sparsectrl/latent_condition.yaml (simplified: one 3x3 conv on VAE latent + mask, :181-184) or
    configs/sparsectrl/image_condition.yaml (SparseControlNetConditioningEmbedding on pixels + mask, :49-82,185-190)."""
    ch = cfg["block_out_channels"]
    temb, xdim, L = ch[0] * 4, cfg["cross_attention_dim"], cfg["layers_per_block"]
    s = OrderedDict()
    s["conv_in.weight"] = (ch[0], cfg["in_channels"], 3, 3)
    s["conv_in.bias"] = (ch[0],)
    if simplified:
        s["controlnet_cond_embedding.weight"] = (ch[0], conditioning_channels + 1, 3, 3)
        s["controlnet_cond_embedding.bias"] = (ch[0],)
    else:
        e = tuple(embedding_channels)
        s["controlnet_cond_embedding.conv_in.weight"] = (e[0], conditioning_channels + 1, 3, 3)
        s["controlnet_cond_embedding.conv_in.bias"] = (e[0],)
        for i in range(len(e) - 1):
            s["controlnet_cond_embedding.blocks.%d.weight" % (2 * i)] = (e[i], e[i], 3, 3)
            s["controlnet_cond_embedding.blocks.%d.bias" % (2 * i)] = (e[i],)
            s["controlnet_cond_embedding.blocks.%d.weight" % (2 * i + 1)] = (e[i + 1], e[i], 3, 3)
            s["controlnet_cond_embedding.blocks.%d.bias" % (2 * i + 1)] = (e[i + 1],)
        s["controlnet_cond_embedding.conv_out.weight"] = (ch[0], e[-1], 3, 3)
        s["controlnet_cond_embedding.conv_out.bias"] = (ch[0],)
    s["time_embedding.linear_1.weight"] = (temb, ch[0])
    s["time_embedding.linear_1.bias"] = (temb,)
    s["time_embedding.linear_2.weight"] = (temb, temb)
    s["time_embedding.linear_2.bias"] = (temb,)
    n_out = 0

    def zero_conv(c):
        nonlocal n_out
        s["controlnet_down_blocks.%d.weight" % n_out] = (c, c, 1, 1)
        s["controlnet_down_blocks.%d.bias" % n_out] = (c,)
        n_out += 1
    zero_conv(ch[0])
    out = ch[0]
    for i in range(4):
        cin, out = out, ch[i]
        for j in range(L):
            s.update(_resnet_shapes("down_blocks.%d.resnets.%d." % (i, j), cin if j == 0 else out, out, temb))
            if cfg["down_has_attn"][i]:
                s.update(_spatial_shapes("down_blocks.%d.attentions.%d." % (i, j), out, xdim))
            mm = _motion_shapes("down_blocks.%d.motion_modules.%d." % (i, j), out)
            s.update(OrderedDict((k, v) for k, v in mm.items() if "attention_blocks.1." not in k and "norms.1." not in k))
            zero_conv(out)
        if i < 3:
            s["down_blocks.%d.downsamplers.0.conv.weight" % i] = (out, out, 3, 3)
            s["down_blocks.%d.downsamplers.0.conv.bias" % i] = (out,)
            zero_conv(out)
    c = ch[-1]
    s.update(_resnet_shapes("mid_block.resnets.0.", c, c, temb))
    s.update(_spatial_shapes("mid_block.attentions.0.", c, xdim))
    s.update(_resnet_shapes("mid_block.resnets.1.", c, c, temb))
    s["controlnet_mid_block.weight"] = (c, c, 1, 1)
    s["controlnet_mid_block.bias"] = (c,)
    return s


def random_controlnet_state_dict(cfg, seed=4321, conditioning_channels=4, simplified=True,
                                 embedding_channels=COND_EMBEDDING_CHANNELS):
    """Seeded synthetic SparseCtrl weights; the zero-initialised layers (cond embedding, output convs, motion
    proj_out) get small random values instead, otherwise the encoder's output would be identically zero."""
    g = torch.Generator().manual_seed(seed)
    shapes = controlnet_param_shapes(cfg, conditioning_channels, simplified, embedding_channels)
    sd = OrderedDict()
    for name, shape in shapes.items():
        if "norm" in name and len(shape) == 1:
            t = (torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)) + 0.05 * torch.randn(shape, generator=g)
        elif "temporal_transformer.proj_out" in name:
            t = 0.02 * torch.randn(shape, generator=g)
        else:
            wshape = shapes[name[:-4] + "weight"] if name.endswith("bias") else shape
            fan_in = 1
            for d in wshape[1:]:
                fan_in *= d
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
        sd[name] = t
    return sd


def motion_module_single(sd, p, x, cfg):
    """the SparseCtrl flavour of VanillaTemporalModule: attention_block_types = ['Temporal_Self'] (one attention)"""
    B, C, F, H, W = x.shape
    heads = cfg["motion_heads"]
    p = p + "temporal_transformer."
    h = _gn(sd, p + "norm.", x, cfg["norm_num_groups"], 1e-6)
    tok = _lin(sd, p + "proj_in.", h.permute(0, 2, 3, 4, 1).reshape(B * F, H * W, C))
    b = p + "transformer_blocks.0."
    pe = temporal_pe(cfg["motion_pe_max_len"], C, x.device).to(x.dtype)
    N = H * W
    n = _ln(sd, b + "norms.0.", tok)
    seq = n.reshape(B, F, N, C).permute(0, 2, 1, 3).reshape(B * N, F, C) + pe[None, :F]
    ap = b + "attention_blocks.0."
    o = _lin(sd, ap + "to_out.0.", _mha(_lin(sd, ap + "to_q.", seq), _lin(sd, ap + "to_k.", seq), _lin(sd, ap + "to_v.", seq), heads))
    tok = o.reshape(B, N, F, C).permute(0, 2, 1, 3).reshape(B * F, N, C) + tok
    tok = _feed_forward(sd, b + "ff.", _ln(sd, b + "ff_norm.", tok)) + tok
    tok = _lin(sd, p + "proj_out.", tok)
    return tok.reshape(B, F, H, W, C).permute(0, 4, 1, 2, 3) + x


def cond_embedding_pyramid(sd, x, p="controlnet_cond_embedding."):
    """SparseControlNetConditioningEmbedding.forward (sparse_controlnet.py:72-82): conv_in, SiLU, then pairs of
    (3x3 same-width, 3x3 stride-2 widening) convs each followed by SiLU, conv_out without activation.
    x [B, channels + 1, F, 8H, 8W] in pixel space -> [B, C0, F, H, W]."""
    h = Fn.silu(_conv(sd, p + "conv_in.", x))
    i = 0
    while (p + "blocks.%d.weight" % i) in sd:
        h = Fn.silu(_conv(sd, p + "blocks.%d." % i, h, stride=2 if i % 2 else 1))
        i += 1
    return _conv(sd, p + "conv_out.", h)


def controlnet_forward(sd, cfg, sample_shape, timestep, text, cond, mask, conditioning_scale=1.0):
    """SparseControlNetModel.forward (sparse_controlnet.py:450-587) with set_noisy_sample_input_to_zero = True:
    returns (12 down residuals, mid residual), each scaled.  The condition embedding is the simplified conv
    (cond = VAE latents, latent resolution) or the pixel-space pyramid, whichever the state-dict holds."""
    B, _, F, H, W = sample_shape
    L = cfg["layers_per_block"]
    t = torch.as_tensor(timestep, device=text.device).reshape(-1).expand(B)
    temb = timestep_embedding(sd, t, cfg["block_out_channels"][0], text.dtype)
    x = sd["conv_in.bias"].reshape(1, -1, 1, 1, 1).expand(B, -1, F, H, W)                     # :516-518
    if "controlnet_cond_embedding.weight" in sd:
        emb = _conv({"weight": sd["controlnet_cond_embedding.weight"], "bias": sd["controlnet_cond_embedding.bias"]}, "",
                    torch.cat([cond, mask], dim=1))                                           # :522-525
    else:
        emb = cond_embedding_pyramid(sd, torch.cat([cond, mask], dim=1))
    x = x + emb
    feats = [x]
    for i in range(4):
        for j in range(L):
            x = resnet_block(sd, "down_blocks.%d.resnets.%d." % (i, j), x, temb, cfg)
            if cfg["down_has_attn"][i]:
                x = spatial_transformer(sd, "down_blocks.%d.attentions.%d." % (i, j), x, text, cfg)
            x = motion_module_single(sd, "down_blocks.%d.motion_modules.%d." % (i, j), x, cfg)
            feats.append(x)
        if i < 3:
            x = _conv(sd, "down_blocks.%d.downsamplers.0.conv." % i, x, stride=2)
            feats.append(x)
    x = resnet_block(sd, "mid_block.resnets.0.", x, temb, cfg)
    x = spatial_transformer(sd, "mid_block.attentions.0.", x, text, cfg)
    x = resnet_block(sd, "mid_block.resnets.1.", x, temb, cfg)
    down = [_conv(sd, "controlnet_down_blocks.%d." % i, f, pad=0) * conditioning_scale for i, f in enumerate(feats)]
    mid = _conv(sd, "controlnet_mid_block.", x, pad=0) * conditioning_scale
    return down, mid
