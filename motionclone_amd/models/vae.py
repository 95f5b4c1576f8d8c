"""Drop-in `AutoencoderKL` for the decode side of the pipeline (`self.vae.decode(z).sample`, reference
motionclone/pipelines/pipeline_animation.py:249-263), backed by the HIP decoder engine.

The reference takes this class from `diffusers==0.16.0` (t2v_video_sample.py:4,25); the mirror keeps that version's
surface as far as the reference touches it: `from_pretrained(path, subfolder="vae")` reading `config.json` +
`diffusion_pytorch_model.bin`, diffusers-style `.config` (incl. `scaling_factor`, `block_out_channels`, which
`AnimationPipeline.__init__` reads for `vae_scale_factor`), `.dtype`, `.device`, `.to()`, `state_dict()` /
`load_state_dict()` under the 0.16.0 key names, `decode(z, return_dict=True)` returning an object with `.sample`
in [N, 3, H, W].  `decode_video` is the batched path `decode_latents` uses when it finds it.

`encode(x).latent_dist.sample()` (motionclone_functions.py:64,125) runs the HIP encoder engine the same way; the normal
draw of the posterior comes from torch's (global) generator exactly where the reference draws it.
"""
import json
import os
from dataclasses import dataclass

import torch
from torch import nn

from ..vae_engine import SD15_VAE_CONFIG, VaeDecoderEngine, VaeEncoderEngine
from .unet import FrozenConfig, ParamNode


@dataclass
class DecoderOutput:
    sample: torch.Tensor


@dataclass
class AutoencoderKLOutput:
    latent_dist: object


def _resnet_shapes(p, cin, cout):
    s = {p + "norm1.weight": (cin,), p + "norm1.bias": (cin,), p + "conv1.weight": (cout, cin, 3, 3),
         p + "conv1.bias": (cout,), p + "norm2.weight": (cout,), p + "norm2.bias": (cout,),
         p + "conv2.weight": (cout, cout, 3, 3), p + "conv2.bias": (cout,)}
    if cin != cout:
        s[p + "conv_shortcut.weight"] = (cout, cin, 1, 1)
        s[p + "conv_shortcut.bias"] = (cout,)
    return s


def _mid_shapes(p, c):
    s = _resnet_shapes(p + "resnets.0.", c, c)
    a = p + "attentions.0."
    s[a + "group_norm.weight"] = (c,)
    s[a + "group_norm.bias"] = (c,)
    for n in ("query", "key", "value", "proj_attn"):
        s[a + n + ".weight"] = (c, c)
        s[a + n + ".bias"] = (c,)
    s.update(_resnet_shapes(p + "resnets.1.", c, c))
    return s


def vae_param_shapes(cfg):
    """diffusers 0.16.0 AutoencoderKL state_dict inventory (encoder, quant_conv, post_quant_conv, decoder)"""
    ch, L, lat = tuple(cfg["block_out_channels"]), cfg["layers_per_block"], cfg["latent_channels"]
    s = {"encoder.conv_in.weight": (ch[0], cfg["in_channels"], 3, 3), "encoder.conv_in.bias": (ch[0],)}
    prev = ch[0]
    for i, c in enumerate(ch):
        for j in range(L):
            s.update(_resnet_shapes("encoder.down_blocks.%d.resnets.%d." % (i, j), prev if j == 0 else c, c))
        if i != len(ch) - 1:
            s["encoder.down_blocks.%d.downsamplers.0.conv.weight" % i] = (c, c, 3, 3)
            s["encoder.down_blocks.%d.downsamplers.0.conv.bias" % i] = (c,)
        prev = c
    s.update(_mid_shapes("encoder.mid_block.", ch[-1]))
    s.update({"encoder.conv_norm_out.weight": (ch[-1],), "encoder.conv_norm_out.bias": (ch[-1],),
              "encoder.conv_out.weight": (2 * lat, ch[-1], 3, 3), "encoder.conv_out.bias": (2 * lat,),
              "quant_conv.weight": (2 * lat, 2 * lat, 1, 1), "quant_conv.bias": (2 * lat,),
              "post_quant_conv.weight": (lat, lat, 1, 1), "post_quant_conv.bias": (lat,),
              "decoder.conv_in.weight": (ch[-1], lat, 3, 3), "decoder.conv_in.bias": (ch[-1],)})
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
    s.update({"decoder.conv_norm_out.weight": (ch[0],), "decoder.conv_norm_out.bias": (ch[0],),
              "decoder.conv_out.weight": (cfg["out_channels"], ch[0], 3, 3), "decoder.conv_out.bias": (cfg["out_channels"],)})
    return s


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 latent_channels=4, norm_num_groups=32, sample_size=512, scaling_factor=0.18215, act_fn="silu",
                 down_block_types=None, up_block_types=None, **unused):
        super().__init__()
        if act_fn != "silu":
            raise NotImplementedError("only the SiLU VAE of SD-1.5 is built")
        n = len(block_out_channels)
        if (down_block_types and tuple(down_block_types) != ("DownEncoderBlock2D",) * n) or \
                (up_block_types and tuple(up_block_types) != ("UpDecoderBlock2D",) * n):
            raise NotImplementedError("only DownEncoderBlock2D / UpDecoderBlock2D stacks are built")
        self.engine_config = dict(SD15_VAE_CONFIG, in_channels=in_channels, out_channels=out_channels,
                                  block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                  latent_channels=latent_channels, norm_num_groups=norm_num_groups,
                                  scaling_factor=scaling_factor)
        self.config = FrozenConfig(in_channels=in_channels, out_channels=out_channels,
                                   block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                   latent_channels=latent_channels, norm_num_groups=norm_num_groups,
                                   sample_size=sample_size, scaling_factor=scaling_factor, act_fn=act_fn)
        for name, shape in vae_param_shapes(self.engine_config).items():
            node = self
            parts = name.split(".")
            for part in parts[:-1]:
                if part not in node._modules:
                    node.add_module(part, ParamNode())
                node = node._modules[part]
            node.register_parameter(parts[-1], nn.Parameter(torch.empty(shape, dtype=torch.float16), requires_grad=False))
        self._engine = None
        self._engine_key = None
        self._enc = None
        self._enc_key = None

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, **kw):
        if subfolder is not None:
            pretrained_model_path = os.path.join(pretrained_model_path, subfolder)
        config_file = os.path.join(pretrained_model_path, "config.json")
        if not os.path.isfile(config_file):
            raise RuntimeError(f"{config_file} does not exist")
        with open(config_file, "r") as f:
            config = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        model = cls(**config)
        model_file = os.path.join(pretrained_model_path, "diffusion_pytorch_model.bin")
        if not os.path.isfile(model_file):
            raise RuntimeError(f"{model_file} does not exist")
        from .. import checkpoints
        model.load_state_dict(checkpoints.read(model_file))
        return model

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._engine = self._enc = None
        return out

    def engine(self):
        p0 = next(self.parameters())
        key = (p0.device, p0.data_ptr())
        if self._engine is None or self._engine_key != key:
            sd = {k: v for k, v in self.state_dict().items() if k.startswith("decoder.") or k.startswith("post_quant_conv.")}
            self._engine = VaeDecoderEngine(sd, self.engine_config, p0.device)
            self._engine_key = key
        return self._engine

    def decode(self, z, return_dict=True):
        sample = self.engine().decode(z)
        return DecoderOutput(sample=sample) if return_dict else (sample,)

    def decode_video(self, latents):
        """[1, 4, F, h, w] -> float32 [1, 3, F, H, W] in [0, 1]: decode_latents without the per-frame loop"""
        return self.engine().decode_video(latents)

    def encoder_engine(self):
        p0 = next(self.parameters())
        key = (p0.device, p0.data_ptr())
        if self._enc is None or self._enc_key != key:
            sd = {k: v for k, v in self.state_dict().items() if k.startswith("encoder.") or k.startswith("quant_conv.")}
            self._enc = VaeEncoderEngine(sd, self.engine_config, p0.device)
            self._enc_key = key
        return self._enc

    def encode(self, x, return_dict=True):
        dist = self.encoder_engine().encode(x)
        return AutoencoderKLOutput(latent_dist=dist) if return_dict else (dist,)

    def forward(self, *a, **k):
        raise NotImplementedError("call decode(); the training-style forward is not part of the sampling path")
