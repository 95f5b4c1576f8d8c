"""ORACLE - TEST INFRASTRUCTURE ONLY.  Never imported by motionclone_amd / motionclone.

CPU (plain PyTorch, fp32) restatement of the VAE the reference calls around its denoising loop:
`AnimationPipeline.decode_latents` (reference motionclone/pipelines/pipeline_animation.py:249-263: latents / 0.18215,
frame-by-frame `self.vae.decode(z).sample`, `(x / 2 + 0.5).clamp(0, 1)`) and the encode side of
`obtain_motion_representation` / `sample_video` (motionclone/utils/motionclone_functions.py:31,64-65,125:
`self.vae.encode(x).latent_dist.sample() * 0.18215`).

The VAE itself is NOT reference source: it is `diffusers==0.16.0` `AutoencoderKL` (pinned at the reference's
environment.yaml:13, absent from /root/reference and from this image).  **Parity unpinned**: what follows restates the
published 0.16.0 algorithm (models/autoencoder_kl.py, models/vae.py: Encoder / Decoder / DiagonalGaussianDistribution,
models/unet_2d_blocks.py: DownEncoderBlock2D / UpDecoderBlock2D / UNetMidBlock2D, models/resnet.py: ResnetBlock2D /
Upsample2D / Downsample2D, models/attention.py: AttentionBlock) over a flat state-dict with that version's key names
(`decoder.mid_block.attentions.0.{group_norm,query,key,value,proj_attn}` ...), and is anchored on the reference's call
sites above.  If a GPU box ever carries real diffusers, assert equality there.

SD-1.5 `vae/config.json`: block_out_channels (128, 256, 512, 512), layers_per_block 2, latent_channels 4,
norm_num_groups 32, act_fn silu, scaling_factor 0.18215.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as Fn

SD15_VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                       layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215)
TINY_VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(64, 128),
                       layers_per_block=1, norm_num_groups=32, scaling_factor=0.18215)
EPS = 1e-6   # resnet_eps passed by Encoder / Decoder to every block, and to conv_norm_out (vae.py)


# ---- parameter inventory (diffusers 0.16.0 AutoencoderKL state_dict keys) ---------------------------
def _resnet_shapes(p, cin, cout):
    s = OrderedDict()
    s[p + "norm1.weight"] = (cin,)
    s[p + "norm1.bias"] = (cin,)
    s[p + "conv1.weight"] = (cout, cin, 3, 3)
    s[p + "conv1.bias"] = (cout,)
    s[p + "norm2.weight"] = (cout,)
    s[p + "norm2.bias"] = (cout,)
    s[p + "conv2.weight"] = (cout, cout, 3, 3)
    s[p + "conv2.bias"] = (cout,)
    if cin != cout:
        s[p + "conv_shortcut.weight"] = (cout, cin, 1, 1)
        s[p + "conv_shortcut.bias"] = (cout,)
    return s


def _mid_shapes(p, c):
    s = OrderedDict()
    s.update(_resnet_shapes(p + "resnets.0.", c, c))
    a = p + "attentions.0."
    s[a + "group_norm.weight"] = (c,)
    s[a + "group_norm.bias"] = (c,)
    for n in ("query", "key", "value", "proj_attn"):
        s[a + n + ".weight"] = (c, c)
        s[a + n + ".bias"] = (c,)
    s.update(_resnet_shapes(p + "resnets.1.", c, c))
    return s


def decoder_param_shapes(cfg):
    ch = tuple(cfg["block_out_channels"])
    L, lat = cfg["layers_per_block"], cfg["latent_channels"]
    s = OrderedDict()
    s["post_quant_conv.weight"] = (lat, lat, 1, 1)
    s["post_quant_conv.bias"] = (lat,)
    s["decoder.conv_in.weight"] = (ch[-1], lat, 3, 3)
    s["decoder.conv_in.bias"] = (ch[-1],)
    s.update(_mid_shapes("decoder.mid_block.", ch[-1]))
    rev = ch[::-1]
    prev = rev[0]
    for i, c in enumerate(rev):
        for j in range(L + 1):
            s.update(_resnet_shapes("decoder.up_blocks.%d.resnets.%d." % (i, j), prev if j == 0 else c, c))
        if i != len(rev) - 1:
            s["decoder.up_blocks.%d.upsamplers.0.conv.weight" % i] = (c, c, 3, 3)
            s["decoder.up_blocks.%d.upsamplers.0.conv.bias" % i] = (c,)
        prev = c
    s["decoder.conv_norm_out.weight"] = (ch[0],)
    s["decoder.conv_norm_out.bias"] = (ch[0],)
    s["decoder.conv_out.weight"] = (cfg["out_channels"], ch[0], 3, 3)
    s["decoder.conv_out.bias"] = (cfg["out_channels"],)
    return s


def encoder_param_shapes(cfg):
    ch = tuple(cfg["block_out_channels"])
    L, lat = cfg["layers_per_block"], cfg["latent_channels"]
    s = OrderedDict()
    s["encoder.conv_in.weight"] = (ch[0], cfg["in_channels"], 3, 3)
    s["encoder.conv_in.bias"] = (ch[0],)
    prev = ch[0]
    for i, c in enumerate(ch):
        for j in range(L):
            s.update(_resnet_shapes("encoder.down_blocks.%d.resnets.%d." % (i, j), prev if j == 0 else c, c))
        if i != len(ch) - 1:
            s["encoder.down_blocks.%d.downsamplers.0.conv.weight" % i] = (c, c, 3, 3)
            s["encoder.down_blocks.%d.downsamplers.0.conv.bias" % i] = (c,)
        prev = c
    s.update(_mid_shapes("encoder.mid_block.", ch[-1]))
    s["encoder.conv_norm_out.weight"] = (ch[-1],)
    s["encoder.conv_norm_out.bias"] = (ch[-1],)
    s["encoder.conv_out.weight"] = (2 * lat, ch[-1], 3, 3)
    s["encoder.conv_out.bias"] = (2 * lat,)
    s["quant_conv.weight"] = (2 * lat, 2 * lat, 1, 1)
    s["quant_conv.bias"] = (2 * lat,)
    return s


def param_shapes(cfg):
    s = encoder_param_shapes(cfg)
    s.update(decoder_param_shapes(cfg))
    return s


def random_state_dict(cfg, seed=4242, dtype=torch.float32):
    """Seeded synthetic weights with PyTorch-default-like scales (kaiming-uniform fan-in bound), norms near identity."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, shape in param_shapes(cfg).items():
        if "norm" in name:
            base = 1.0 if name.endswith("weight") else 0.0
            sd[name] = (base + 0.1 * torch.randn(shape, generator=g)).to(dtype)
        elif name.endswith("bias"):
            sd[name] = (0.02 * torch.randn(shape, generator=g)).to(dtype)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            bound = 1.0 / math.sqrt(fan_in)
            sd[name] = ((torch.rand(shape, generator=g) * 2 - 1) * bound).to(dtype)
    return sd


# ---- building blocks --------------------------------------------------------------------------------
def _conv(sd, p, x, stride=1, pad=1):
    return Fn.conv2d(x, sd[p + "weight"], sd[p + "bias"], stride=stride, padding=pad)


def _gn(sd, p, x, groups):
    return Fn.group_norm(x, groups, sd[p + "weight"], sd[p + "bias"], EPS)


def resnet_block(sd, p, x, groups):
    """ResnetBlock2D without time embedding (diffusers 0.16.0 models/resnet.py), output_scale_factor = 1"""
    h = _conv(sd, p + "conv1.", Fn.silu(_gn(sd, p + "norm1.", x, groups)))
    h = _conv(sd, p + "conv2.", Fn.silu(_gn(sd, p + "norm2.", h, groups)))
    if p + "conv_shortcut.weight" in sd:
        x = _conv(sd, p + "conv_shortcut.", x, pad=0)
    return x + h


def attention_block(sd, p, x, groups):
    """AttentionBlock, one head (attn_num_head_channels=None): softmax(q k^T / sqrt(C)) v in fp32 (models/attention.py)"""
    b, c, h, w = x.shape
    n = _gn(sd, p + "group_norm.", x, groups).reshape(b, c, h * w).transpose(1, 2)
    q = Fn.linear(n, sd[p + "query.weight"], sd[p + "query.bias"])
    k = Fn.linear(n, sd[p + "key.weight"], sd[p + "key.bias"])
    v = Fn.linear(n, sd[p + "value.weight"], sd[p + "value.bias"])
    probs = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * (1.0 / math.sqrt(c)), dim=-1)
    o = Fn.linear(torch.bmm(probs, v), sd[p + "proj_attn.weight"], sd[p + "proj_attn.bias"])
    return o.transpose(1, 2).reshape(b, c, h, w) + x


def mid_block(sd, p, x, groups):
    x = resnet_block(sd, p + "resnets.0.", x, groups)
    x = attention_block(sd, p + "attentions.0.", x, groups)
    return resnet_block(sd, p + "resnets.1.", x, groups)


# ---- decode / encode ---------------------------------------------------------------------------------
def decode(sd, cfg, z):
    """AutoencoderKL.decode: post_quant_conv -> Decoder.forward.  z [N, latent, h, w] -> [N, 3, 8h, 8w]"""
    G, L = cfg["norm_num_groups"], cfg["layers_per_block"]
    n_up = len(cfg["block_out_channels"])
    x = _conv(sd, "post_quant_conv.", z, pad=0)
    x = _conv(sd, "decoder.conv_in.", x)
    x = mid_block(sd, "decoder.mid_block.", x, G)
    for i in range(n_up):
        for j in range(L + 1):
            x = resnet_block(sd, "decoder.up_blocks.%d.resnets.%d." % (i, j), x, G)
        if i != n_up - 1:   # Upsample2D: nearest 2x then conv
            x = _conv(sd, "decoder.up_blocks.%d.upsamplers.0.conv." % i, Fn.interpolate(x, scale_factor=2.0, mode="nearest"))
    x = Fn.silu(_gn(sd, "decoder.conv_norm_out.", x, G))
    return _conv(sd, "decoder.conv_out.", x)


def decode_latents(sd, cfg, latents):
    """reference pipeline_animation.py:249-263.  latents [1, 4, F, h, w] -> video float32 [1, 3, F, H, W] in [0, 1]"""
    b, c, f, h, w = latents.shape
    z = (1.0 / 0.18215) * latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    frames = torch.cat([decode(sd, cfg, z[i:i + 1]) for i in range(z.shape[0])])
    video = frames.reshape(b, f, -1, frames.shape[-2], frames.shape[-1]).permute(0, 2, 1, 3, 4)
    return (video / 2 + 0.5).clamp(0, 1).float()


def encode_moments(sd, cfg, x):
    """AutoencoderKL.encode up to the posterior parameters.  x [N, 3, H, W] in [-1, 1] -> (mean, std) [N, latent, H/8, W/8]"""
    G, L = cfg["norm_num_groups"], cfg["layers_per_block"]
    n_down = len(cfg["block_out_channels"])
    h = _conv(sd, "encoder.conv_in.", x)
    for i in range(n_down):
        for j in range(L):
            h = resnet_block(sd, "encoder.down_blocks.%d.resnets.%d." % (i, j), h, G)
        if i != n_down - 1:   # Downsample2D(padding=0): pad right / bottom by one, stride-2 conv without padding
            h = _conv(sd, "encoder.down_blocks.%d.downsamplers.0.conv." % i, Fn.pad(h, (0, 1, 0, 1)), stride=2, pad=0)
    h = mid_block(sd, "encoder.mid_block.", h, G)
    h = _conv(sd, "encoder.conv_out.", Fn.silu(_gn(sd, "encoder.conv_norm_out.", h, G)))
    moments = _conv(sd, "quant_conv.", h, pad=0)
    mean, logvar = moments.chunk(2, dim=1)
    return mean, torch.exp(0.5 * logvar.clamp(-30.0, 20.0))


def encode_sample(sd, cfg, x, noise):
    """latent_dist.sample() with the normal draw passed in (the reference draws it from the global RNG)"""
    mean, std = encode_moments(sd, cfg, x)
    return mean + std * noise
