"""Parameter inventory of the reference UNet3DConditionModel (state-dict names and shapes) and seeded
synthetic weights for measurement.  Names are a compatibility contract: SD-1.5 diffusers UNet keys plus
AnimateDiff `...motion_modules.N.temporal_transformer...` keys (reference unet.py:477-515, util.py:130-137).
"""
import math
from collections import OrderedDict

import torch


def _res(s, p, cin, cout, temb):
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


def _attn(s, p, c, kv):
    s[p + "to_q.weight"] = (c, c)
    s[p + "to_k.weight"] = (c, kv)
    s[p + "to_v.weight"] = (c, kv)
    s[p + "to_out.0.weight"] = (c, c)
    s[p + "to_out.0.bias"] = (c,)


def _ff(s, p, c):
    s[p + "net.0.proj.weight"] = (8 * c, c)
    s[p + "net.0.proj.bias"] = (8 * c,)
    s[p + "net.2.weight"] = (c, 4 * c)
    s[p + "net.2.bias"] = (c,)


def _norm(s, p, c):
    s[p + "weight"] = (c,)
    s[p + "bias"] = (c,)


def _spatial(s, p, c, xdim):
    _norm(s, p + "norm.", c)
    s[p + "proj_in.weight"] = (c, c, 1, 1)
    s[p + "proj_in.bias"] = (c,)
    b = p + "transformer_blocks.0."
    _attn(s, b + "attn1.", c, c)
    _norm(s, b + "norm1.", c)
    _attn(s, b + "attn2.", c, xdim)
    _norm(s, b + "norm2.", c)
    _ff(s, b + "ff.", c)
    _norm(s, b + "norm3.", c)
    s[p + "proj_out.weight"] = (c, c, 1, 1)
    s[p + "proj_out.bias"] = (c,)


def _motion(s, p, c):
    p += "temporal_transformer."
    _norm(s, p + "norm.", c)
    s[p + "proj_in.weight"] = (c, c)
    s[p + "proj_in.bias"] = (c,)
    b = p + "transformer_blocks.0."
    for a in range(2):
        _attn(s, b + "attention_blocks.%d." % a, c, c)
    for a in range(2):
        _norm(s, b + "norms.%d." % a, c)
    _ff(s, b + "ff.", c)
    _norm(s, b + "ff_norm.", c)
    s[p + "proj_out.weight"] = (c, c)
    s[p + "proj_out.bias"] = (c,)


def param_shapes(cfg):
    ch = cfg["block_out_channels"]
    temb, xdim, L = ch[0] * 4, cfg["cross_attention_dim"], cfg["layers_per_block"]
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
            _res(s, "down_blocks.%d.resnets.%d." % (i, j), cin if j == 0 else out, out, temb)
            if cfg["down_has_attn"][i]:
                _spatial(s, "down_blocks.%d.attentions.%d." % (i, j), out, xdim)
            _motion(s, "down_blocks.%d.motion_modules.%d." % (i, j), out)
        if i < 3:
            s["down_blocks.%d.downsamplers.0.conv.weight" % i] = (out, out, 3, 3)
            s["down_blocks.%d.downsamplers.0.conv.bias" % i] = (out,)
    c = ch[-1]
    _res(s, "mid_block.resnets.0.", c, c, temb)
    _spatial(s, "mid_block.attentions.0.", c, xdim)
    _res(s, "mid_block.resnets.1.", c, c, temb)
    rev = list(reversed(ch))
    out = rev[0]
    for i in range(4):
        prev, out = out, rev[i]
        skip_in = rev[min(i + 1, 3)]
        for j in range(L + 1):
            skip = skip_in if j == L else out
            rin = prev if j == 0 else out
            _res(s, "up_blocks.%d.resnets.%d." % (i, j), rin + skip, out, temb)
            if cfg["up_has_attn"][i]:
                _spatial(s, "up_blocks.%d.attentions.%d." % (i, j), out, xdim)
            _motion(s, "up_blocks.%d.motion_modules.%d." % (i, j), out)
        if i < 3:
            s["up_blocks.%d.upsamplers.0.conv.weight" % i] = (out, out, 3, 3)
            s["up_blocks.%d.upsamplers.0.conv.bias" % i] = (out,)
    _norm(s, "conv_norm_out.", ch[0])
    s["conv_out.weight"] = (cfg["out_channels"], ch[0], 3, 3)
    s["conv_out.bias"] = (cfg["out_channels"],)
    return s


def controlnet_param_shapes(cfg, conditioning_channels=4, simplified=True, embedding_channels=(16, 32, 96, 256)):
    """reference SparseControlNetModel keys (sparse_controlnet.py:150-314): `simplified` = the one-conv latent condition
    embedding of latent_condition.yaml (:181-184), else the pixel-space SparseControlNetConditioningEmbedding of
    image_condition.yaml (:49-82,185-190; `conditioning_channels` counts the image channels, the mask adds one)"""
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
    outs = [ch[0]]
    out = ch[0]
    for i in range(4):
        cin, out = out, ch[i]
        for j in range(L):
            _res(s, "down_blocks.%d.resnets.%d." % (i, j), cin if j == 0 else out, out, temb)
            if cfg["down_has_attn"][i]:
                _spatial(s, "down_blocks.%d.attentions.%d." % (i, j), out, xdim)
            mm = OrderedDict()
            _motion(mm, "down_blocks.%d.motion_modules.%d." % (i, j), out)
            s.update((k, v) for k, v in mm.items() if "attention_blocks.1." not in k and "norms.1." not in k)
            outs.append(out)
        if i < 3:
            s["down_blocks.%d.downsamplers.0.conv.weight" % i] = (out, out, 3, 3)
            s["down_blocks.%d.downsamplers.0.conv.bias" % i] = (out,)
            outs.append(out)
    for i, c in enumerate(outs):
        s["controlnet_down_blocks.%d.weight" % i] = (c, c, 1, 1)
        s["controlnet_down_blocks.%d.bias" % i] = (c,)
    c = ch[-1]
    _res(s, "mid_block.resnets.0.", c, c, temb)
    _spatial(s, "mid_block.attentions.0.", c, xdim)
    _res(s, "mid_block.resnets.1.", c, c, temb)
    s["controlnet_mid_block.weight"] = (c, c, 1, 1)
    s["controlnet_mid_block.bias"] = (c,)
    return s


def synthetic_controlnet_state_dict(cfg, seed=4321, device="cuda", dtype=torch.float16):
    """seeded random SparseCtrl weights (zero-initialised layers get small random values, see synthetic_state_dict)"""
    shapes = controlnet_param_shapes(cfg)
    g = torch.Generator(device=device).manual_seed(seed)
    sd = OrderedDict()
    for name, shape in shapes.items():
        if len(shape) == 1 and "norm" in name:
            t = (1.0 if name.endswith("weight") else 0.0) + 0.05 * torch.randn(shape, generator=g, device=device)
        elif "temporal_transformer.proj_out" in name:
            t = 0.02 * torch.randn(shape, generator=g, device=device)
        else:
            wshape = shapes[name[:-4] + "weight"] if name.endswith("bias") else shape
            fan_in = 1
            for d in wshape[1:]:
                fan_in *= d
            t = (torch.rand(shape, generator=g, device=device) * 2 - 1) / math.sqrt(fan_in)
        sd[name] = t.to(dtype)
    return sd


def synthetic_state_dict(cfg, seed=1234, device="cuda", dtype=torch.float16, flat=None):
    """Seeded random-init weights of the named architecture (no checkpoints exist offline; BASELINE.json asks for
    synthetic data).  fan-in-uniform conv/linear weights as PyTorch's default init, unit norms, and motion
    `proj_out` ~ N(0, 0.02) instead of the zero init of motion_module.py:77-78 (a zero proj_out would remove the
    temporal path from eps).  `flat` (optional) is a preallocated 1-D buffer the tensors are carved from, so a
    multi-GPU launch can broadcast all weights with a single RCCL call."""
    shapes = param_shapes(cfg)
    total = sum(int(torch.Size(s).numel()) for s in shapes.values())
    if flat is None:
        flat = torch.empty(total, dtype=dtype, device=device)
    assert flat.numel() >= total
    g = torch.Generator(device=flat.device).manual_seed(seed)
    sd = OrderedDict()
    off = 0
    for name, shape in shapes.items():
        n = int(torch.Size(shape).numel())
        t = flat[off:off + n].view(shape)
        off += n
        if len(shape) == 1 and "norm" in name:
            t.copy_((1.0 if name.endswith("weight") else 0.0) + 0.05 * torch.randn(shape, generator=g, device=flat.device))
        elif "temporal_transformer.proj_out" in name:
            t.copy_(0.02 * torch.randn(shape, generator=g, device=flat.device))
        else:
            wshape = shapes[name[:-4] + "weight"] if name.endswith("bias") else shape
            fan_in = 1
            for d in wshape[1:]:
                fan_in *= d
            t.copy_((torch.rand(shape, generator=g, device=flat.device) * 2 - 1) / math.sqrt(fan_in))
        sd[name] = t
    return sd, flat
