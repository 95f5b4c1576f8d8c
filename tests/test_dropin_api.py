"""The drop-in boundary: the reference's public names, state-dict keys and patch points, driven the
way t2v_video_sample.py:36-73 drives them, produce the same latents as the oracle's loop."""
import subprocess
import sys
import types

import torch

from motionclone_amd.models.unet import UNet3DConditionModel
from motionclone_amd.pipelines.pipeline_animation import AnimationPipeline
from motionclone_amd.scheduler import DDIMSchedulerState
from motionclone_amd.utils import motionclone_functions as mf
from oracle import guidance_ref as G
from oracle import unet3d_ref as U


def build_pipeline(dev, cfg, sd, N, Gs, gscale):
    unet = UNet3DConditionModel(in_channels=4, out_channels=4, block_out_channels=cfg["block_out_channels"],
                                layers_per_block=2, cross_attention_dim=cfg["cross_attention_dim"],
                                attention_head_dim=cfg["attention_heads"], use_motion_module=True,
                                motion_module_resolutions=[1, 2, 4, 8], motion_module_mid_block=False,
                                motion_module_kwargs=dict(num_attention_heads=cfg["motion_heads"],
                                                          num_transformer_block=1,
                                                          attention_block_types=["Temporal_Self", "Temporal_Self"],
                                                          temporal_position_encoding=True))
    assert set(unet.state_dict().keys()) == set(sd.keys())
    missing, unexpected = unet.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    unet = unet.to(dev).to(dtype=torch.float16)
    pipeline = AnimationPipeline(vae=None, text_encoder=None, tokenizer=None, unet=unet, controlnet=None,
                                 scheduler=DDIMSchedulerState(beta_start=0.00085, beta_end=0.012,
                                                              beta_schedule="linear", steps_offset=1, clip_sample=False))
    # t2v_video_sample.py:57-65
    pipeline.scheduler.customized_step = mf.schedule_customized_step.__get__(pipeline.scheduler)
    pipeline.scheduler.customized_set_timesteps = mf.schedule_set_timesteps.__get__(pipeline.scheduler)
    pipeline.unet.forward = mf.unet_customized_forward.__get__(pipeline.unet)
    pipeline.sample_video = mf.sample_video.__get__(pipeline)
    pipeline.single_step_video = mf.single_step_video.__get__(pipeline)
    pipeline.get_temp_attn_prob = mf.get_temp_attn_prob.__get__(pipeline)
    pipeline.add_noise = mf.add_noise.__get__(pipeline)
    pipeline.compute_temp_loss = mf.compute_temp_loss.__get__(pipeline)
    pipeline.obtain_motion_representation = mf.obtain_motion_representation.__get__(pipeline)
    for p in pipeline.unet.parameters():
        p.requires_grad = False
    config = types.SimpleNamespace(cfg_scale=7.5, negative_prompt="", inference_steps=N, guidance_scale=gscale,
                                   guidance_steps=Gs, warm_up_steps=10, cool_up_steps=10, motion_guidance_weight=2000,
                                   motion_guidance_blocks=["up_blocks.1"], add_noise_step=400, video_length=4,
                                   height=64, width=64, new_prompt="x")
    pipeline.input_config, pipeline.unet.input_config = config, config
    pipeline.unet = mf.prep_unet_attention(pipeline.unet, config.motion_guidance_blocks)
    pipeline.unet = mf.prep_unet_conv(pipeline.unet)
    pipeline.scheduler.customized_set_timesteps(N, Gs, gscale, device=dev, timestep_spacing_type="uneven")
    return pipeline


def test_entry_script_flow_matches_oracle(backend):
    dev = backend
    cfg = dict(U.TINY_CONFIG)
    sd = {k: v.half().float() for k, v in U.random_state_dict(cfg, seed=1234).items()}
    N, Gs, gscale = 3, 2, 0.3
    pipe = build_pipeline(dev, cfg, sd, N, Gs, gscale)
    hooked = [n for n, m in pipe.unet.named_modules() if getattr(m, "processor", None) is not None]
    assert len(hooked) == 6 and all("up_blocks.1.motion_modules" in n for n in hooked)
    assert len(pipe.unet.up_blocks[1].resnets) == 3 and pipe.unet.config.in_channels == 4

    text = torch.randn(2, 7, cfg["cross_attention_dim"], generator=torch.Generator().manual_seed(7)).half()
    vid = (0.18215 * torch.randn(1, 4, 4, 8, 8, generator=torch.Generator().manual_seed(11))).half()
    gen = torch.Generator(device=dev).manual_seed(5)
    rep = pipe.obtain_motion_representation(generator=gen, motion_representation_path=None,
                                            video_latents=vid.to(dev), uncond_embeddings=text[0:1].to(dev))
    prob = pipe.get_temp_attn_prob()
    assert list(prob) == list(rep) and prob[hooked[0]].shape == (4, cfg["motion_heads"], 4, 4)
    # the representation is the top-1 of those probabilities (motionclone_functions.py:79)
    v, i = torch.topk(prob[hooked[0]].float(), 1, -1)
    assert (v.cpu() - rep[hooked[0]][0].float().cpu()).abs().max() < 2e-3
    loss = pipe.compute_temp_loss(prob)
    assert loss.item() < 1e-5  # same latents, same representation -> zero guidance loss
    # index_select (:267-271): 0/1 flags over equal blocks of the (b n) axis
    half = pipe.get_temp_attn_prob(index_select=[0, 1])
    assert all(torch.equal(half[k], prob[k][prob[k].shape[0] // 2:]) for k in prob)
    quarter = pipe.get_temp_attn_prob(index_select=[1, 0, 0, 1])
    n4 = prob[hooked[0]].shape[0] // 4
    assert torch.equal(quarter[hooked[0]], torch.cat([prob[hooked[0]][:n4], prob[hooked[0]][3 * n4:]]))

    lat0 = torch.randn(1, 4, 4, 8, 8, generator=torch.Generator().manual_seed(2025)).half()
    out = pipe.sample_video(generator=None, noisy_latents=lat0.to(dev), text_embeddings=text.to(dev), decode=False)
    hp = dict(cfg_scale=7.5, motion_guidance_weight=2000.0, warm_up_steps=10, cool_up_steps=10)
    rep_cpu = {k: [a.float().cpu(), b.cpu()] for k, (a, b) in rep.items()}
    ref = G.sample_loop(sd, cfg, lat0.float(), text.float(), rep_cpu, hp, N, Gs, gscale)
    err = ((out.float().cpu() - ref).norm() / ref.norm()).item()
    assert err < 3e-2, err

    # standalone scheduler step (schedule_customized_step) against the oracle's DDIM restatement
    eps = torch.randn(1, 4, 4, 8, 8, generator=torch.Generator().manual_seed(3)).half()
    score = torch.randn(1, 4, 4, 8, 8, generator=torch.Generator().manual_seed(4))
    got = pipe.scheduler.customized_step(eps.to(dev), 0, lat0.to(dev), score=score.to(dev), return_dict=False)[0]
    want = G.ddim_step(G.alphas_cumprod(), G.uneven_timesteps(N, Gs, gscale), 0, eps.float(), lat0.float(), score)
    assert ((got.float().cpu() - want).norm() / want.norm()).item() < 5e-3


def test_motionclone_alias_package_exports_reference_names():
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from motionclone.models.unet import UNet3DConditionModel\n"
            "from motionclone.pipelines.pipeline_animation import AnimationPipeline\n"
            "from motionclone.utils.util import load_weights\n"
            "from motionclone.utils.motionclone_functions import *\n"
            "from motionclone.utils.xformer_attention import *\n"
            "for n in ('schedule_customized_step schedule_set_timesteps unet_customized_forward sample_video "
            "single_step_video get_temp_attn_prob add_noise compute_temp_loss obtain_motion_representation "
            "prep_unet_attention prep_unet_conv set_all_seed os np').split():\n"
            "    assert n in globals(), n\n"
            "print('ok')\n") % __import__("os").path.dirname(__import__("os").path.dirname(__file__))
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr


def test_i2v_flow_with_sparsectrl_matches_oracle(backend):
    """i2v_video_sample.py flow: SparseControlNetModel.from_unet + add_controlnet sampling vs the oracle loop"""
    from motionclone_amd.models.sparse_controlnet import SparseControlNetModel
    dev = backend
    cfg = dict(U.TINY_CONFIG)
    sd = {k: v.half().float() for k, v in U.random_state_dict(cfg, seed=1234).items()}
    csd = {k: v.half().float() for k, v in U.random_controlnet_state_dict(cfg).items()}
    N, Gs, gscale = 3, 2, 0.3
    pipe = build_pipeline(dev, cfg, sd, N, Gs, gscale)
    ckw = dict(set_noisy_sample_input_to_zero=True, use_simplified_condition_embedding=True, conditioning_channels=4,
               use_motion_module=True, motion_module_resolutions=[1, 2, 4, 8], motion_module_mid_block=False,
               motion_module_type="Vanilla",
               motion_module_kwargs=dict(num_attention_heads=cfg["motion_heads"], num_transformer_block=1,
                                         attention_block_types=["Temporal_Self"], temporal_position_encoding=True,
                                         temporal_position_encoding_max_len=32, temporal_attention_dim_div=1))
    controlnet = SparseControlNetModel.from_unet(pipe.unet, controlnet_additional_kwargs=ckw)
    assert set(controlnet.state_dict().keys()) == set(csd.keys())
    controlnet.load_state_dict(csd)
    pipe.controlnet = controlnet.to(dev).to(dtype=torch.float16)
    pipe.input_config.image_index = [0]
    pipe.input_config.controlnet_scale = 0.8

    text = torch.randn(2, 7, cfg["cross_attention_dim"], generator=torch.Generator().manual_seed(7)).half()
    vid = (0.18215 * torch.randn(1, 4, 4, 8, 8, generator=torch.Generator().manual_seed(11))).half()
    img_lat = (0.18215 * torch.randn(1, 4, 1, 8, 8, generator=torch.Generator().manual_seed(12))).half()
    rep = pipe.obtain_motion_representation(generator=torch.Generator(device=dev).manual_seed(5), use_controlnet=True,
                                            video_latents=vid.to(dev), uncond_embeddings=text[0:1].to(dev))
    lat0 = torch.randn(1, 4, 4, 8, 8, generator=torch.Generator().manual_seed(2025)).half()
    out = pipe.sample_video(noisy_latents=lat0.to(dev), text_embeddings=text.to(dev), decode=False, add_controlnet=True,
                            controlnet_images=img_lat.to(dev))
    # oracle loop with the same representation and per-step encoder residuals
    rep_cpu = {k: [a.float().cpu(), b.cpu()] for k, (a, b) in rep.items()}
    cond = torch.zeros(1, 4, 4, 8, 8)
    mask = torch.zeros(1, 1, 4, 8, 8)
    cond[:, :, [0]] = img_lat.float()
    mask[:, :, [0]] = 1
    ts = G.uneven_timesteps(N, Gs, gscale)
    hp = dict(cfg_scale=7.5, motion_guidance_weight=2000.0, warm_up_steps=10, cool_up_steps=10, guidance_steps=Gs)
    x = lat0.float()
    for i in range(N):
        with torch.no_grad():
            d, m = U.controlnet_forward(csd, cfg, (2, 4, 4, 8, 8), int(ts[i]), text.float(), cond, mask, 0.8)
        if i < Gs:
            x, _ = G.guided_step(sd, cfg, x, i, ts, text.float(), rep_cpu, hp, res_u=([t[[0]] for t in d], m[[0]]),
                                 res_c=([t[[1]] for t in d], m[[1]]))
        else:
            x, _ = G.plain_step_full(sd, cfg, x, i, ts, text.float(), 7.5, res=(d, m))
    err = ((out.float().cpu() - x).norm() / x.norm()).item()
    assert err < 3e-2, err
