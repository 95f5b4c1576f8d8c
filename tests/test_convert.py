"""Checkpoint conversion / LoRA merge: motionclone_amd.utils.convert against the reference's
own converters (loaded by file path from /root/reference, skipped where the tree is absent) and against an inverse
mapping written independently here (round trip - this half also runs on the GPU box)."""
import importlib.util
import os
import re
import types

import pytest
import torch

from motionclone_amd.utils import convert as C
from oracle import reference_shim as shim
from oracle import unet3d_ref as U
from oracle import vae_ref as V


def _ref(name):
    shim.install()
    path = os.path.join(shim.REFERENCE_ROOT, "motionclone", "utils", name + ".py")
    spec = importlib.util.spec_from_file_location("_ref_" + name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


needs_ref = pytest.mark.skipif(not shim.available(), reason="reference tree not present")


# ---- independent inverse maps: diffusers layout -> LDM layout --------------------------------------------
def unet_to_ldm(sd, L=2, n_levels=4):
    res = {"norm1": "in_layers.0", "conv1": "in_layers.2", "time_emb_proj": "emb_layers.1", "norm2": "out_layers.0",
           "conv2": "out_layers.3", "conv_shortcut": "skip_connection"}
    out = {}
    for k, v in sd.items():
        if "motion_modules" in k:
            continue
        m = re.match(r"(down|up)_blocks\.(\d+)\.(resnets|attentions)\.(\d+)\.(.+)", k)
        if m:
            side, lvl, kind, j, leaf = m.group(1), int(m.group(2)), m.group(3), int(m.group(4)), m.group(5)
            n = (1 + lvl * (L + 1) + j) if side == "down" else (lvl * (L + 1) + j)
            stem = ("input_blocks.%d." if side == "down" else "output_blocks.%d.") % n
            if kind == "resnets":
                head, _, tail = leaf.partition(".")
                out["model.diffusion_model." + stem + "0." + res[head] + "." + tail] = v
            else:
                out["model.diffusion_model." + stem + "1." + leaf] = v
            continue
        m = re.match(r"down_blocks\.(\d+)\.downsamplers\.0\.conv\.(.+)", k)
        if m:
            n = 1 + int(m.group(1)) * (L + 1) + L
            out["model.diffusion_model.input_blocks.%d.0.op.%s" % (n, m.group(2))] = v
            continue
        m = re.match(r"up_blocks\.(\d+)\.upsamplers\.0\.conv\.(.+)", k)
        if m:
            lvl = int(m.group(1))
            n = lvl * (L + 1) + L
            has_attn = any(kk.startswith("up_blocks.%d.attentions." % lvl) for kk in sd)
            out["model.diffusion_model.output_blocks.%d.%d.conv.%s" % (n, 2 if has_attn else 1, m.group(2))] = v
            continue
        m = re.match(r"mid_block\.(resnets|attentions)\.(\d)\.(.+)", k)
        if m:
            idx = {"resnets0": 0, "attentions0": 1, "resnets1": 2}[m.group(1) + m.group(2)]
            leaf = m.group(3)
            if m.group(1) == "resnets":
                head, _, tail = leaf.partition(".")
                leaf = res[head] + "." + tail
            out["model.diffusion_model.middle_block.%d.%s" % (idx, leaf)] = v
            continue
        for a, b in (("time_embedding.linear_1.", "time_embed.0."), ("time_embedding.linear_2.", "time_embed.2."),
                     ("conv_in.", "input_blocks.0.0."), ("conv_norm_out.", "out.0."), ("conv_out.", "out.2.")):
            if k.startswith(a):
                out["model.diffusion_model." + b + k[len(a):]] = v
    return out


def vae_to_ldm(sd, n_levels):
    att = {"group_norm": "norm", "query": "q", "key": "k", "value": "v", "proj_attn": "proj_out"}
    out = {}
    for k, v in sd.items():
        side, _, rest = k.partition(".")
        if side in ("quant_conv", "post_quant_conv"):
            out["first_stage_model." + k] = v
            continue
        m = re.match(r"(down|up)_blocks\.(\d+)\.resnets\.(\d+)\.(.+)", rest)
        if m:
            lvl = int(m.group(2)) if m.group(1) == "down" else n_levels - 1 - int(m.group(2))
            out["first_stage_model.%s.%s.%d.block.%s.%s" % (side, m.group(1), lvl, m.group(3),
                                                           m.group(4).replace("conv_shortcut", "nin_shortcut"))] = v
            continue
        m = re.match(r"down_blocks\.(\d+)\.downsamplers\.0\.conv\.(.+)", rest)
        if m:
            out["first_stage_model.%s.down.%s.downsample.conv.%s" % (side, m.group(1), m.group(2))] = v
            continue
        m = re.match(r"up_blocks\.(\d+)\.upsamplers\.0\.conv\.(.+)", rest)
        if m:
            out["first_stage_model.%s.up.%d.upsample.conv.%s" % (side, n_levels - 1 - int(m.group(1)), m.group(2))] = v
            continue
        m = re.match(r"mid_block\.resnets\.(\d)\.(.+)", rest)
        if m:
            out["first_stage_model.%s.mid.block_%d.%s" % (side, int(m.group(1)) + 1, m.group(2))] = v
            continue
        m = re.match(r"mid_block\.attentions\.0\.(\w+)\.(weight|bias)", rest)
        if m:
            t = v[:, :, None, None] if (m.group(2) == "weight" and m.group(1) != "group_norm") else v
            out["first_stage_model.%s.mid.attn_1.%s.%s" % (side, att[m.group(1)], m.group(2))] = t
            continue
        if rest.startswith("conv_norm_out."):
            out["first_stage_model.%s.norm_out.%s" % (side, rest[len("conv_norm_out."):])] = v
        else:
            out["first_stage_model." + k] = v
    return out


def _same(a, b):
    assert set(a) == set(b), (sorted(set(a) - set(b))[:5], sorted(set(b) - set(a))[:5])
    for k in a:
        assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k


@pytest.fixture(scope="module")
def unet_sd():
    sd = U.random_state_dict(U.TINY_CONFIG, seed=3)
    return {k: v for k, v in sd.items() if "motion_modules" not in k}


def test_unet_round_trip(unet_sd):
    ldm = unet_to_ldm(unet_sd)
    ldm["model_ema.decay"] = torch.zeros(())           # ignored
    ldm["cond_stage_model.transformer.text_model.x"] = torch.ones(2)
    assert len(ldm) == len(unet_sd) + 2
    _same(C.convert_ldm_unet_checkpoint(ldm, dict(layers_per_block=2)), unet_sd)
    assert C.convert_ldm_clip_checkpoint_concise(ldm) == {"text_model.x": ldm["cond_stage_model.transformer.text_model.x"]}


def test_vae_round_trip():
    for cfg in (V.TINY_VAE_CONFIG, V.SD15_VAE_CONFIG):
        shapes = V.param_shapes(cfg)
        sd = {k: torch.randn(s) if len(s) < 3 or s[0] * s[1] < 70000 else torch.zeros(s) for k, s in shapes.items()}
        _same(C.convert_ldm_vae_checkpoint(vae_to_ldm(sd, len(cfg["block_out_channels"]))), sd)


@needs_ref
def test_converters_match_reference(unet_sd):
    R = _ref("convert_from_ckpt")
    ldm = unet_to_ldm(unet_sd)
    cfg = dict(layers_per_block=2, class_embed_type=None)
    _same(C.convert_ldm_unet_checkpoint(dict(ldm), cfg), R.convert_ldm_unet_checkpoint(dict(ldm), cfg))
    vcfg = V.SD15_VAE_CONFIG
    vsd = {k: torch.randn(s) if len(s) < 3 else torch.zeros(s) for k, s in V.param_shapes(vcfg).items()}
    vldm = vae_to_ldm(vsd, 4)
    ref_cfg = dict(down_block_types=["DownEncoderBlock2D"] * 4, up_block_types=["UpDecoderBlock2D"] * 4)
    _same(C.convert_ldm_vae_checkpoint(dict(vldm), ref_cfg), R.convert_ldm_vae_checkpoint(dict(vldm), ref_cfg))
    clip = {"cond_stage_model.transformer.text_model.embeddings.token_embedding.weight": torch.ones(3, 2),
            "model.diffusion_model.out.2.bias": torch.zeros(4)}
    _same(C.convert_ldm_clip_checkpoint_concise(clip), R.convert_ldm_clip_checkpoint_concise(clip))


class _Lin(torch.nn.Module):
    def __init__(self, o, i, conv=False):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(o, i, 1, 1) if conv else torch.randn(o, i))


def _toy_unet():
    root = torch.nn.Module()
    blk = torch.nn.Module()
    attn = torch.nn.Module()
    attn.to_q = _Lin(6, 4)
    attn.to_out = torch.nn.ModuleList([_Lin(4, 6)])
    blk.attn1 = attn
    blk.proj_in = _Lin(4, 4, conv=True)
    root.down_blocks = torch.nn.ModuleList([blk])
    return root


def _lora_sd(seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    kohya = {"lora_unet_down_blocks_0_attn1_to_q.lora_down.weight": r(2, 4), "lora_unet_down_blocks_0_attn1_to_q.lora_up.weight": r(6, 2),
             "lora_unet_down_blocks_0_attn1_to_q.alpha": torch.tensor(2.0),
             "lora_unet_down_blocks_0_proj_in.lora_down.weight": r(2, 4, 1, 1), "lora_unet_down_blocks_0_proj_in.lora_up.weight": r(4, 2, 1, 1)}
    diff = {"down_blocks.0.attn1.processor.to_q_lora.down.weight": r(2, 4), "down_blocks.0.attn1.processor.to_q_lora.up.weight": r(6, 2),
            "down_blocks.0.attn1.processor.to_out_lora.down.weight": r(2, 6), "down_blocks.0.attn1.processor.to_out_lora.up.weight": r(4, 2)}
    return kohya, diff


def test_lora_merge_math():
    kohya, diff = _lora_sd(0)
    u = _toy_unet()
    w0 = {k: v.detach().clone() for k, v in u.state_dict().items()}
    pipe = types.SimpleNamespace(unet=u, text_encoder=None)
    C.convert_lora(pipe, kohya, alpha=0.7)
    d = 0.7 * kohya["lora_unet_down_blocks_0_attn1_to_q.lora_up.weight"] @ kohya["lora_unet_down_blocks_0_attn1_to_q.lora_down.weight"]
    assert torch.allclose(u.down_blocks[0].attn1.to_q.weight, w0["down_blocks.0.attn1.to_q.weight"] + d, atol=1e-6)
    assert not torch.equal(u.down_blocks[0].proj_in.weight, w0["down_blocks.0.proj_in.weight"])
    C.load_diffusers_lora(pipe, diff, alpha=0.5)
    d2 = 0.5 * diff["down_blocks.0.attn1.processor.to_out_lora.up.weight"] @ diff["down_blocks.0.attn1.processor.to_out_lora.down.weight"]
    assert torch.allclose(u.down_blocks[0].attn1.to_out[0].weight, w0["down_blocks.0.attn1.to_out.0.weight"] + d2, atol=1e-6)
    with pytest.raises(KeyError):
        C.convert_lora(pipe, {"lora_unet_nope.lora_down.weight": torch.zeros(1, 1), "lora_unet_nope.lora_up.weight": torch.zeros(1, 1)})


@needs_ref
def test_lora_merge_matches_reference():
    R = _ref("convert_lora_safetensor_to_diffusers")
    kohya, diff = _lora_sd(1)
    a, b = _toy_unet(), _toy_unet()
    b.load_state_dict(a.state_dict())
    pa, pb = types.SimpleNamespace(unet=a, text_encoder=None), types.SimpleNamespace(unet=b, text_encoder=None)
    C.convert_lora(pa, kohya, alpha=0.8)
    R.convert_lora(pb, kohya, alpha=0.8)
    C.load_diffusers_lora(pa, diff, alpha=1.0)
    R.load_diffusers_lora(pb, diff, alpha=1.0)
    for (k, x), (_, y) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.allclose(x, y, atol=1e-6), k


def test_load_weights_end_to_end(tmp_path):
    """load_weights with an original-layout checkpoint (UNet + VAE), a kohya LoRA and a motion module on the native
    model mirrors: weights land under the diffusers / AnimateDiff names, engines are invalidated"""
    from safetensors.torch import save_file
    from motionclone_amd.models.unet import UNet3DConditionModel
    from motionclone_amd.models.vae import AutoencoderKL
    from motionclone_amd.utils.util import load_weights
    cfg = U.TINY_CONFIG
    full = U.random_state_dict(cfg, seed=9)
    sd2d = {k: v for k, v in full.items() if "motion_modules" not in k}
    vcfg = V.TINY_VAE_CONFIG
    vsd = V.random_state_dict(vcfg, seed=10)
    ckpt = unet_to_ldm(sd2d)
    ckpt.update(vae_to_ldm(vsd, len(vcfg["block_out_channels"])))
    save_file({k: v.contiguous() for k, v in ckpt.items()}, str(tmp_path / "db.safetensors"))
    torch.save({"state_dict": {k: v for k, v in full.items() if "motion_modules" in k}}, str(tmp_path / "mm.ckpt"))
    unet = UNet3DConditionModel(block_out_channels=cfg["block_out_channels"], cross_attention_dim=cfg["cross_attention_dim"],
                                attention_head_dim=cfg["attention_heads"],
                                motion_module_kwargs=dict(num_attention_heads=cfg["motion_heads"]))
    vae = AutoencoderKL(block_out_channels=vcfg["block_out_channels"], layers_per_block=vcfg["layers_per_block"])
    pipe = types.SimpleNamespace(unet=unet, vae=vae, text_encoder=None)
    key = "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight"
    r = torch.randn(2, full[key].shape[1]), torch.randn(full[key].shape[0], 2)
    lora = {"lora_unet_" + key[:-7].replace(".", "_") + ".lora_down.weight": r[0],
            "lora_unet_" + key[:-7].replace(".", "_") + ".lora_up.weight": r[1]}
    save_file(lora, str(tmp_path / "l.safetensors"))
    load_weights(pipe, motion_module_path=str(tmp_path / "mm.ckpt"), dreambooth_model_path=str(tmp_path / "db.safetensors"),
                 lora_model_path=str(tmp_path / "l.safetensors"), lora_alpha=0.5)
    got = unet.state_dict()
    for k, v in full.items():
        want = v + 0.5 * (r[1] @ r[0]) if k == key else v
        assert torch.allclose(got[k].float(), want, atol=2e-3, rtol=2e-3), k
    for k, v in vsd.items():
        assert torch.allclose(vae.state_dict()[k].float(), v, atol=2e-3, rtol=2e-3), k
