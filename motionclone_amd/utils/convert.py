"""Checkpoint key conversion and LoRA merging for `load_weights` (reference motionclone/utils/util.py:115-215, which
calls `convert_ldm_unet_checkpoint` / `convert_ldm_vae_checkpoint` / `convert_ldm_clip_checkpoint_concise`
(motionclone/utils/convert_from_ckpt.py:328,559,716), `convert_lora` and `load_diffusers_lora`
(motionclone/utils/convert_lora_safetensor_to_diffusers.py:50,27)).  SURVEY.md 8(f) rank 2: load-time host code.

Written as data: an original-Stable-Diffusion ("LDM") checkpoint differs from the diffusers layout only in how blocks
are numbered and in a handful of leaf names, so each converter is a prefix strip + one pass of `_rename` with a
structural index map derived from the keys themselves (which sub-module of `input_blocks.N` / `output_blocks.N`
is a resnet, an attention or a resampler is read off its leaf names, not assumed from a config).
`tests/test_convert.py` checks every converter against the reference's own functions on synthetic checkpoints.
"""
import re

import torch

UNET_PREFIX = "model.diffusion_model."
VAE_PREFIX = "first_stage_model."
CLIP_PREFIX = "cond_stage_model.transformer."

_RESNET_LEAVES = (("in_layers.0.", "norm1."), ("in_layers.2.", "conv1."), ("emb_layers.1.", "time_emb_proj."),
                  ("out_layers.0.", "norm2."), ("out_layers.3.", "conv2."), ("skip_connection.", "conv_shortcut."))


def _strip(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def _submodule_kinds(keys, stem):
    """{sub-index: 'resnet' | 'attn' | 'down' | 'up'} for the children of `stem.N.` found in `keys`"""
    kinds = {}
    pat = re.compile(re.escape(stem) + r"(\d+)\.(\d+)\.(.+)")
    for k in keys:
        m = pat.match(k)
        if not m:
            continue
        n, sub, leaf = int(m.group(1)), int(m.group(2)), m.group(3)
        if leaf.startswith("in_layers."):
            kind = "resnet"
        elif leaf.startswith("op."):
            kind = "down"
        elif leaf.startswith("conv."):
            kind = "up"
        elif leaf.startswith(("transformer_blocks.", "proj_in.", "proj_out.", "norm.")):
            kind = "attn"
        else:
            continue
        kinds.setdefault(n, {})[sub] = kind
    return kinds


def convert_ldm_unet_checkpoint(checkpoint, config=None, path=None, extract_ema=False, controlnet=False):
    """LDM UNet keys -> diffusers UNet2DCondition keys (same call shape as convert_from_ckpt.py:328; `config` only
    supplies layers_per_block, default 2; EMA weights are not extracted, as with the reference's default)."""
    if extract_ema or controlnet:
        raise NotImplementedError("EMA extraction / ControlNet checkpoints are not used by load_weights")
    src = _strip(checkpoint, UNET_PREFIX)
    L = int((config or {}).get("layers_per_block", 2)) if isinstance(config, dict) else int(getattr(config, "layers_per_block", 2))
    out = {}
    for a, b in (("time_embed.0.", "time_embedding.linear_1."), ("time_embed.2.", "time_embedding.linear_2."),
                 ("input_blocks.0.0.", "conv_in."), ("out.0.", "conv_norm_out."), ("out.2.", "conv_out.")):
        for k, v in src.items():
            if k.startswith(a):
                out[b + k[len(a):]] = v

    def put_resnet(dst, stem):
        for k, v in src.items():
            if k.startswith(stem):
                leaf = k[len(stem):]
                for a, b in _RESNET_LEAVES:
                    if leaf.startswith(a):
                        out[dst + b + leaf[len(a):]] = v

    def put_same(dst, stem):
        for k, v in src.items():
            if k.startswith(stem):
                out[dst + k[len(stem):]] = v

    for n, subs in sorted(_submodule_kinds(src, "input_blocks.").items()):
        if n == 0:
            continue
        level, j = (n - 1) // (L + 1), (n - 1) % (L + 1)
        for sub, kind in subs.items():
            stem = "input_blocks.%d.%d." % (n, sub)
            if kind == "resnet":
                put_resnet("down_blocks.%d.resnets.%d." % (level, j), stem)
            elif kind == "attn":
                put_same("down_blocks.%d.attentions.%d." % (level, j), stem)
            elif kind == "down":
                put_same("down_blocks.%d.downsamplers.0.conv." % level, stem + "op.")
    put_resnet("mid_block.resnets.0.", "middle_block.0.")
    put_same("mid_block.attentions.0.", "middle_block.1.")
    put_resnet("mid_block.resnets.1.", "middle_block.2.")
    for n, subs in sorted(_submodule_kinds(src, "output_blocks.").items()):
        level, j = n // (L + 1), n % (L + 1)
        for sub, kind in subs.items():
            stem = "output_blocks.%d.%d." % (n, sub)
            if kind == "resnet":
                put_resnet("up_blocks.%d.resnets.%d." % (level, j), stem)
            elif kind == "attn":
                put_same("up_blocks.%d.attentions.%d." % (level, j), stem)
            elif kind == "up":
                put_same("up_blocks.%d.upsamplers.0.conv." % level, stem + "conv.")
    return out


_VAE_ATTN = (("norm.", "group_norm."), ("q.", "query."), ("k.", "key."), ("v.", "value."), ("proj_out.", "proj_attn."))


def convert_ldm_vae_checkpoint(checkpoint, config=None):
    """LDM first-stage keys -> diffusers 0.16.0 AutoencoderKL keys (convert_from_ckpt.py:559).  Attention 1x1 convs become
    Linear weights [C, C]; decoder `up.i` is numbered from the output side, diffusers' `up_blocks` from the input."""
    src = _strip(checkpoint, VAE_PREFIX)
    n_up = 1 + max([int(m.group(1)) for m in (re.match(r"decoder\.up\.(\d+)\.", k) for k in src) if m] or [0])
    out = {}
    for k, v in src.items():
        side, _, rest = k.partition(".")
        if side in ("quant_conv", "post_quant_conv"):
            out[k] = v
            continue
        if side not in ("encoder", "decoder"):
            continue
        m = re.match(r"(down|up)\.(\d+)\.(block\.(\d+)|downsample\.conv|upsample\.conv)\.(.+)", rest)
        if m:
            lvl = int(m.group(2))
            blk = "down_blocks.%d." % lvl if m.group(1) == "down" else "up_blocks.%d." % (n_up - 1 - lvl)
            if m.group(3).startswith("block"):
                leaf = m.group(5).replace("nin_shortcut.", "conv_shortcut.")
                out["%s.%sresnets.%s.%s" % (side, blk, m.group(4), leaf)] = v
            elif m.group(3) == "downsample.conv":
                out["%s.%sdownsamplers.0.conv.%s" % (side, blk, m.group(5))] = v
            else:
                out["%s.%supsamplers.0.conv.%s" % (side, blk, m.group(5))] = v
            continue
        m = re.match(r"mid\.block_(\d)\.(.+)", rest)
        if m:
            out["%s.mid_block.resnets.%d.%s" % (side, int(m.group(1)) - 1, m.group(2).replace("nin_shortcut.", "conv_shortcut."))] = v
            continue
        m = re.match(r"mid\.attn_1\.(.+)", rest)
        if m:
            leaf = m.group(1)
            for a, b in _VAE_ATTN:
                if leaf.startswith(a):
                    t = v
                    if leaf.endswith("weight") and a != "norm." and t.dim() == 4:
                        t = t[:, :, 0, 0]
                    out["%s.mid_block.attentions.0.%s%s" % (side, b, leaf[len(a):])] = t
            continue
        if rest.startswith("norm_out."):
            out["%s.conv_norm_out.%s" % (side, rest[len("norm_out."):])] = v
        elif rest.startswith(("conv_in.", "conv_out.")):
            out[k] = v
    return out


def convert_ldm_clip_checkpoint_concise(checkpoint):
    """convert_from_ckpt.py:716: the text encoder keys are the HF CLIPTextModel keys behind a prefix"""
    return _strip(checkpoint, CLIP_PREFIX)


# ---- LoRA merging (weights are updated in place, no adapter modules) -----------------------------------------
def _delta(up, down, alpha):
    up, down = up.float(), down.float()
    if up.dim() == 4:
        return (alpha * (up[:, :, 0, 0] @ down[:, :, 0, 0]))[:, :, None, None]
    return alpha * (up @ down)


def _flat_index(module):
    """{'down_blocks_0_attentions_0_..._to_q': parameter} for every `.weight` of `module` (kohya flattens '.' to '_')"""
    return {name[:-len(".weight")].replace(".", "_"): p for name, p in module.named_parameters() if name.endswith(".weight")}


def convert_lora(pipeline, state_dict, LORA_PREFIX_UNET="lora_unet", LORA_PREFIX_TEXT_ENCODER="lora_te", alpha=0.6):
    """kohya-format LoRA (`lora_unet_<path>.lora_up/.lora_down.weight`) merged as W += alpha * up @ down
    (convert_lora_safetensor_to_diffusers.py:50-115; the `.alpha` tensors are ignored there too)."""
    index = {LORA_PREFIX_UNET: _flat_index(pipeline.unet)}
    if getattr(pipeline, "text_encoder", None) is not None:
        index[LORA_PREFIX_TEXT_ENCODER] = _flat_index(pipeline.text_encoder)
    with torch.no_grad():
        for key, down in state_dict.items():
            if ".lora_down." not in key:
                continue
            stem = key.split(".")[0]
            prefix = LORA_PREFIX_TEXT_ENCODER if stem.startswith(LORA_PREFIX_TEXT_ENCODER + "_") else LORA_PREFIX_UNET
            target = index.get(prefix, {}).get(stem[len(prefix) + 1:])
            if target is None:
                raise KeyError("LoRA key %s has no matching layer" % key)
            up = state_dict[key.replace(".lora_down.", ".lora_up.")]
            target.data += _delta(up, down, alpha).to(target.data)
    _invalidate(pipeline)
    return pipeline


def load_diffusers_lora(pipeline, state_dict, alpha=1.0):
    """diffusers-attention-processor-format LoRA (AnimateDiff domain adapter, motion LoRA):
    `<path>.processor.to_q_lora.down.weight` -> `<path>.to_q.weight` (convert_lora_safetensor_to_diffusers.py:27-47)"""
    params = dict(pipeline.unet.named_parameters())
    with torch.no_grad():
        for key, down in state_dict.items():
            if "up." in key:
                continue
            name = key.replace("processor.", "").replace("_lora", "").replace("down.", "").replace("up.", "")
            name = name.replace("to_out.", "to_out.0.")
            if name not in params:
                raise KeyError("LoRA key %s -> %s has no matching layer" % (key, name))
            up = state_dict[key.replace(".down.", ".up.")]
            params[name].data += _delta(up, down, alpha).to(params[name].data)
    _invalidate(pipeline)
    return pipeline


def _invalidate(pipeline):
    """the packed-weight engines are rebuilt on next use"""
    for m in (getattr(pipeline, "unet", None), getattr(pipeline, "vae", None), getattr(pipeline, "controlnet", None)):
        for attr in ("_engine", "_enc"):
            if m is not None and hasattr(m, attr):
                setattr(m, attr, None)
