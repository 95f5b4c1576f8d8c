"""Drop-in `UNet3DConditionModel` (reference motionclone/models/unet.py:38-515) backed by the HIP engine.

Mirrors what the guidance layer and the entry scripts touch: construction via
`from_pretrained_2d`, diffusers-style `.config`, `.dtype`, `.device`, `.to()`, `parameters()`,
`state_dict()` / `load_state_dict(strict=False)` under the reference's key names, `named_modules()`
exposing the temporal attentions as `VersatileAttention` objects with `.heads`, `.processor`,
`.set_processor`, `len(up_blocks[i].resnets)`, and `forward(sample, timestep, encoder_hidden_states, ...,
only_motion_feature)` returning an object with `.sample` in the reference layout [B, C, F, H, W].
The arithmetic never runs in PyTorch: `forward` hands the latent to `UNet3DEngine`.
"""
import json
import os
from dataclasses import dataclass

import torch
from torch import nn

from .. import ops, spec
from ..engine import UNet3DEngine, default_config


class FrozenConfig(dict):
    __getattr__ = dict.__getitem__


@dataclass
class UNet3DConditionOutput:
    sample: torch.Tensor


class ParamNode(nn.Module):
    """Generic container reproducing the reference's module tree (names only)."""

    def __len__(self):
        return sum(1 for k in self._modules if k.isdigit())

    def __getitem__(self, i):
        return self._modules[str(i)]

    def __iter__(self):
        return (self._modules[str(i)] for i in range(len(self)))


class VersatileAttention(ParamNode):
    """Temporal self-attention node (reference motion_module.py:250-345).  The attribute contract used by
    prep_unet_attention / get_temp_attn_prob (xformer_attention.py:45-52, motionclone_functions.py:264-281)."""

    def __init__(self, heads):
        super().__init__()
        self.heads = heads
        self.processor = None
        self.recorded = None  # engine-side record: fused q|k|v token matrix + geometry

    def set_processor(self, processor):
        self.processor = processor

    def reshape_heads_to_batch_dim(self, tensor):
        b, n, dim = tensor.shape
        h = self.heads
        return tensor.reshape(b, n, h, dim // h).permute(0, 2, 1, 3).reshape(b * h, n, dim // h)


class ResnetBlock3D(ParamNode):
    pass


_CLASS_BY_SUFFIX = (("attention_blocks.", VersatileAttention),)


def _build_tree(root, shapes, cfg, dtype):
    for name, shape in shapes.items():
        parts = name.split(".")
        node = root
        for depth, part in enumerate(parts[:-1]):
            if part not in node._modules:
                path = ".".join(parts[:depth + 1])
                if "attention_blocks" in parts and parts[depth - 1] == "attention_blocks" and part.isdigit():
                    child = VersatileAttention(cfg["motion_heads"])
                elif depth >= 1 and parts[depth - 1] == "resnets" and part.isdigit():
                    child = ResnetBlock3D()
                else:
                    child = ParamNode()
                child._mc_path = path
                node.add_module(part, child)
            node = node._modules[part]
        node.register_parameter(parts[-1], nn.Parameter(torch.empty(shape, dtype=dtype), requires_grad=False))


class UNet3DConditionModel(nn.Module):
    def __init__(self, sample_size=None, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                 layers_per_block=2, cross_attention_dim=768, attention_head_dim=8, norm_num_groups=32, norm_eps=1e-5,
                 use_motion_module=True, motion_module_resolutions=(1, 2, 4, 8), motion_module_mid_block=False,
                 motion_module_type="Vanilla", motion_module_kwargs=None, use_inflated_groupnorm=True, **unused):
        super().__init__()
        mk = dict(motion_module_kwargs or {})
        if not use_motion_module or motion_module_mid_block or tuple(motion_module_resolutions) != (1, 2, 4, 8):
            raise NotImplementedError("only the AnimateDiff layout of configs/model_config/model_config.yaml is built")
        if mk.get("attention_block_types", ["Temporal_Self", "Temporal_Self"]) != ["Temporal_Self", "Temporal_Self"] \
                or mk.get("num_transformer_block", 1) != 1 or not mk.get("temporal_position_encoding", True):
            raise NotImplementedError("motion_module_kwargs other than the v3_sd15_mm layout")
        heads = attention_head_dim if isinstance(attention_head_dim, int) else attention_head_dim[0]
        self.engine_config = dict(default_config(), in_channels=in_channels, out_channels=out_channels,
                                  block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                  cross_attention_dim=cross_attention_dim, attention_heads=heads,
                                  norm_num_groups=norm_num_groups, norm_eps=norm_eps,
                                  motion_heads=mk.get("num_attention_heads", 8),
                                  motion_pe_max_len=mk.get("temporal_position_encoding_max_len", 32))
        self.config = FrozenConfig(sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
                                   block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                   cross_attention_dim=cross_attention_dim, attention_head_dim=attention_head_dim,
                                   norm_num_groups=norm_num_groups, norm_eps=norm_eps, center_input_sample=False)
        _build_tree(self, spec.param_shapes(self.engine_config), self.engine_config, torch.float16)
        self.num_upsamplers = 3
        self.input_config = None
        self._engine = None
        self._engine_key = None
        self._weights_version = 0

    # ---- diffusers-style surface ----------------------------------------------------------------------
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def enable_xformers_memory_efficient_attention(self, *a, **k):
        """No-op: spatial attention is always the flash-style HIP kernel (reference attention.py:535-542 slot)."""

    @classmethod
    def from_pretrained_2d(cls, pretrained_model_path, subfolder=None, unet_additional_kwargs=None):
        """reference unet.py:477-515: read the SD `unet/config.json`, inflate to 3D, load the 2D weights non-strictly."""
        if subfolder is not None:
            pretrained_model_path = os.path.join(pretrained_model_path, subfolder)
        config_file = os.path.join(pretrained_model_path, "config.json")
        if not os.path.isfile(config_file):
            raise RuntimeError(f"{config_file} does not exist")
        with open(config_file, "r") as f:
            config = json.load(f)
        keys = ("sample_size", "in_channels", "out_channels", "block_out_channels", "layers_per_block",
                "cross_attention_dim", "attention_head_dim", "norm_num_groups", "norm_eps")
        kw = {k: config[k] for k in keys if k in config}
        kw.update(unet_additional_kwargs or {})
        model = cls(**kw)
        model_file = os.path.join(pretrained_model_path, "diffusion_pytorch_model.bin")
        if not os.path.isfile(model_file):
            raise RuntimeError(f"{model_file} does not exist")
        from .. import checkpoints
        state_dict = checkpoints.read(model_file)
        m, u = model.load_state_dict(state_dict, strict=False)
        print(f"### motion keys will be loaded: {len(m)}; \n### unexpected keys: {len(u)};")
        return model

    def load_state_dict(self, state_dict, strict=True, **kw):
        state_dict = {k: v for k, v in state_dict.items() if "pos_encoder.pe" not in k}
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self.invalidate_engine()
        return out

    # ---- execution --------------------------------------------------------------------------------------
    def engine(self):
        """The packed-weight execution engine for the current parameters/device (rebuilt after weight changes)."""
        p0 = next(self.parameters())
        blocks = ("up_blocks.1",)
        if self.input_config is not None:    # re-assigned per example by the entry scripts (t2v_video_sample.py:82)
            blocks = tuple(self.input_config.motion_guidance_blocks)
        key = (p0.device, p0.data_ptr(), blocks, self._weights_version)
        if self._engine is None or self._engine_key != key:
            sd = {k: v for k, v in self.state_dict().items()}
            self._engine = UNet3DEngine(sd, self.engine_config, p0.device, guidance_blocks=blocks)
            self._engine_key = key
        return self._engine

    def invalidate_engine(self):
        """Call after editing parameters in place (e.g. `weight.data += ...`, as LoRA merges do): the engine packs fp16
        copies of the weights lazily and would otherwise keep serving the old ones.  load_state_dict does this itself."""
        self._weights_version += 1
        self._engine = None

    def temporal_attentions(self):
        return [(n, m) for n, m in self.named_modules() if isinstance(m, VersatileAttention)]

    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, attention_mask=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None, return_dict=True,
                only_motion_feature=False):
        """unet_customized_forward (motionclone_functions.py:478-662).  Inference-only: the guidance gradient is
        produced by `engine().guided_eps_and_grad`, not by autograd through this call."""
        if class_labels is not None or attention_mask is not None:
            raise NotImplementedError("class_labels / attention_mask are never passed on the MotionClone path")
        eng = self.engine()
        t = int(timestep) if not torch.is_tensor(timestep) else int(timestep.reshape(-1)[0].item())
        hooked = {n: m for n, m in self.temporal_attentions() if m.processor is not None}
        record = {} if hooked else None
        def as_tokens(r):   # SparseCtrl residuals: token matrices pass through, reference-layout tensors are converted
            if r is None or r.dim() == 2:
                return r
            if r.dim() == 4:
                r = r.unsqueeze(2).expand(-1, -1, sample.shape[2], -1, -1)
            return ops.latent_to_cl(r.to(torch.float16).contiguous(), r.shape[1])
        if down_block_additional_residuals is not None:
            down_block_additional_residuals = [as_tokens(r) for r in down_block_additional_residuals]
            mid_block_additional_residual = as_tokens(mid_block_additional_residual)
        eps = eng.forward(sample.to(torch.float16), t, encoder_hidden_states.to(torch.float16), record=record,
                          only_motion_feature=only_motion_feature,
                          down_residuals=down_block_additional_residuals, mid_residual=mid_block_additional_residual)
        if record:
            for n, m in hooked.items():
                if n in record:
                    m.recorded = record[n]
                    m.processor.record_qkv(m, None, record[n], record[n], None, None)
        if only_motion_feature:
            return 0
        B, _, F, H, W = sample.shape
        out = ops.cl_to_latent(eps, B, self.config.out_channels, F, H, W)
        if not return_dict:
            return (out,)
        return UNet3DConditionOutput(sample=out)
