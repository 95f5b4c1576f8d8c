"""SparseCtrl (BASELINE config 4): the reference's SparseControlNetModel vs the oracle restatement
(CPU, needs /root/reference), and the HIP ControlNetEngine + residual-injected guided / plain steps vs the oracle."""
import pytest
import torch

from motionclone_amd.engine import ControlNetEngine, UNet3DEngine
from motionclone_amd.sampler import MotionCloneSampler
from oracle import guidance_ref as G
from oracle import reference_shim as shim
from oracle import unet3d_ref as U

HP = dict(cfg_scale=7.5, motion_guidance_weight=2000.0, warm_up_steps=10, cool_up_steps=10)


def cond_inputs(F=4, H=8, W=8):
    cond = torch.zeros(1, 4, F, H, W)
    mask = torch.zeros(1, 1, F, H, W)
    cond[:, :, 0] = (0.18215 * torch.randn(1, 4, H, W, generator=torch.Generator().manual_seed(1)))
    mask[:, :, 0] = 1
    return cond.half().float(), mask


def rel(a, b):
    return ((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm().clamp_min(1e-12)).item()


@pytest.mark.skipif(not shim.available(), reason="reference tree not present")
def test_oracle_controlnet_matches_reference():
    shim.install()
    from motionclone.models.sparse_controlnet import SparseControlNetModel
    cfg = dict(U.TINY_CONFIG)
    ref = SparseControlNetModel(
        in_channels=4, down_block_types=("CrossAttnDownBlock3D",) * 3 + ("DownBlock3D",),
        block_out_channels=cfg["block_out_channels"], layers_per_block=2, cross_attention_dim=cfg["cross_attention_dim"],
        attention_head_dim=cfg["attention_heads"], num_attention_heads=cfg["attention_heads"],
        set_noisy_sample_input_to_zero=True, use_simplified_condition_embedding=True, conditioning_channels=4,
        motion_module_kwargs=dict(num_attention_heads=cfg["motion_heads"], num_transformer_block=1,
                                  attention_block_types=["Temporal_Self"], temporal_position_encoding=True,
                                  temporal_position_encoding_max_len=32, temporal_attention_dim_div=1)).float()
    ref.controlnet_cond_embedding.float()
    shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items() if "pos_encoder" not in k}
    assert shapes == {k: tuple(v) for k, v in U.controlnet_param_shapes(cfg).items()}
    sd = U.random_controlnet_state_dict(cfg)
    ref.load_state_dict(sd, strict=False)
    text = torch.randn(2, 7, cfg["cross_attention_dim"], generator=torch.Generator().manual_seed(7))
    cond, mask = cond_inputs()
    lat = torch.randn(2, 4, 4, 8, 8)
    real_to = torch.Tensor.to

    def keep_fp32(self, *a, **k):   # the reference hard-casts the condition to fp16 (sparse_controlnet.py:523)
        return self if (a and a[0] is torch.float16) else real_to(self, *a, **k)
    torch.Tensor.to = keep_fp32
    try:
        with torch.no_grad():
            d_ref, m_ref = ref(lat, 500, encoder_hidden_states=text, controlnet_cond=cond, conditioning_mask=mask,
                               conditioning_scale=0.7, guess_mode=False, return_dict=False)
    finally:
        torch.Tensor.to = real_to
    with torch.no_grad():
        d, m = U.controlnet_forward(sd, cfg, lat.shape, 500, text, cond, mask, 0.7)
    assert len(d) == 12 and max((a - b).abs().max().item() for a, b in zip(d_ref, d)) < 1e-4
    assert (m_ref - m).abs().max().item() < 1e-4


def test_engine_controlnet_and_conditioned_steps(backend):
    dev = backend
    cfg = dict(U.TINY_CONFIG)
    sd = {k: v.half().float() for k, v in U.random_state_dict(cfg, seed=1234).items()}
    csd = {k: v.half().float() for k, v in U.random_controlnet_state_dict(cfg).items()}
    F, H, W = 4, 8, 8
    lat = torch.randn(1, 4, F, H, W, generator=torch.Generator().manual_seed(2025)).half()
    text = torch.randn(2, 7, cfg["cross_attention_dim"], generator=torch.Generator().manual_seed(7)).half()
    vid = (0.18215 * torch.randn(1, 4, F, H, W, generator=torch.Generator().manual_seed(11))).half()
    noise = torch.randn(1, 4, F, H, W, generator=torch.Generator().manual_seed(3)).half()
    cond, mask = cond_inputs(F, H, W)
    scale = 0.8
    N, Gs, gscale = 4, 2, 0.3
    ts = G.uneven_timesteps(N, Gs, gscale)
    eng = UNet3DEngine(sd, cfg, dev)
    ceng = ControlNetEngine(csd, cfg, dev)
    smp = MotionCloneSampler(eng, num_inference_steps=N, guidance_steps=Gs, guidance_scale=gscale, controlnet=ceng, **HP)
    ctrl = dict(cond=cond.half().to(dev), mask=mask.half().to(dev), scale=scale)

    # encoder output
    down, mid = ceng.forward((2, 4, F, H, W), int(ts[0]), text.to(dev), ctrl["cond"], ctrl["mask"], scale)
    with torch.no_grad():
        d_ref, m_ref = U.controlnet_forward(csd, cfg, (2, 4, F, H, W), int(ts[0]), text.float(), cond, mask, scale)

    def tok(t):  # [B, C, F, H, W] -> tokens
        B, C, F_, H_, W_ = t.shape
        return t.permute(0, 2, 3, 4, 1).reshape(-1, C)
    assert len(down) == 12
    for a, b in zip(down, d_ref):
        assert rel(a, tok(b)) < 2e-2
    assert rel(mid, tok(m_ref)) < 2e-2

    # extraction with the encoder (i2v), then a guided and a plain step with residuals
    rep = smp.extract(vid.to(dev), noise.to(dev), text[0:1].to(dev), ctrl=ctrl)
    noisy = smp.add_noise(400, vid, noise).float()
    with torch.no_grad():
        dr, mr = U.controlnet_forward(csd, cfg, noisy.shape, 400, text[[0]].float(), cond, mask, scale)
        rec = {}
        U.unet_forward(sd, cfg, noisy, 400, text[[0]].float(), only_motion_feature=True, record=rec, down_residuals=dr,
                       mid_residual=mr)
        rep_ref = G.motion_representation(G.temp_attn_prob(rec, cfg["motion_heads"]))
    for k in rep_ref:
        assert (rep[k][0].float().cpu() - rep_ref[k][0]).abs().max() < 5e-3
    hp = dict(HP, guidance_steps=Gs)
    aux = {}
    nxt = smp.step(lat.to(dev), 0, text.to(dev), eng.prepare_representation(rep_ref), aux=aux, ctrl=ctrl)
    with torch.no_grad():
        d2, m2 = U.controlnet_forward(csd, cfg, (2, 4, F, H, W), int(ts[0]), text.float(), cond, mask, scale)
    res_u = ([d[[0]] for d in d2], m2[[0]])
    res_c = ([d[[1]] for d in d2], m2[[1]])
    ref_nxt, ref_aux = G.guided_step(sd, cfg, lat.float(), 0, ts, text.float(), rep_ref, hp, res_u=res_u, res_c=res_c)
    assert rel(aux["grad"], ref_aux["grad"]) < 5e-2
    assert rel(nxt, ref_nxt) < 2e-2
    p = smp.step(nxt, Gs, text.to(dev), eng.prepare_representation(rep_ref), ctrl=ctrl)
    with torch.no_grad():
        d3, m3 = U.controlnet_forward(csd, cfg, (2, 4, F, H, W), int(ts[Gs]), text.float(), cond, mask, scale)
    ref_p, _ = G.plain_step_full(sd, cfg, nxt.float().cpu(), Gs, ts, text.float(), HP["cfg_scale"], res=(d3, m3))
    assert rel(p, ref_p) < 2e-2


# ---- image_condition.yaml: pixel-space condition through SparseControlNetConditioningEmbedding (scribble / sketch) --------
EMB = (16, 32, 96, 256)


def pixel_cond_inputs(F=2, H=64, W=64):
    cond = torch.zeros(1, 3, F, H, W)
    mask = torch.zeros(1, 1, F, H, W)
    cond[:, :, 0] = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(1))     # ToTensor range [0, 1]
    mask[:, :, 0] = 1
    return cond.half().float(), mask


@pytest.mark.skipif(not shim.available(), reason="reference tree not present")
def test_oracle_pixel_condition_controlnet_matches_reference():
    shim.install()
    from motionclone.models.sparse_controlnet import SparseControlNetModel
    cfg = dict(U.TINY_CONFIG)
    ref = SparseControlNetModel(
        in_channels=4, down_block_types=("CrossAttnDownBlock3D",) * 3 + ("DownBlock3D",),
        block_out_channels=cfg["block_out_channels"], layers_per_block=2, cross_attention_dim=cfg["cross_attention_dim"],
        attention_head_dim=cfg["attention_heads"], num_attention_heads=cfg["attention_heads"],
        set_noisy_sample_input_to_zero=True, use_simplified_condition_embedding=False, conditioning_channels=3,
        motion_module_kwargs=dict(num_attention_heads=cfg["motion_heads"], num_transformer_block=1,
                                  attention_block_types=["Temporal_Self"], temporal_position_encoding=True,
                                  temporal_position_encoding_max_len=32, temporal_attention_dim_div=1)).float()
    ref.controlnet_cond_embedding.float()
    shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items() if "pos_encoder" not in k}
    assert shapes == {k: tuple(v) for k, v in U.controlnet_param_shapes(cfg, 3, simplified=False).items()}
    sd = U.random_controlnet_state_dict(cfg, conditioning_channels=3, simplified=False)
    ref.load_state_dict(sd, strict=False)
    text = torch.randn(2, 7, cfg["cross_attention_dim"], generator=torch.Generator().manual_seed(7))
    cond, mask = pixel_cond_inputs()
    lat = torch.randn(2, 4, 2, 8, 8)
    real_to = torch.Tensor.to

    def keep_fp32(self, *a, **k):   # the reference hard-casts the condition to fp16 (sparse_controlnet.py:523)
        return self if (a and a[0] is torch.float16) else real_to(self, *a, **k)
    torch.Tensor.to = keep_fp32
    try:
        with torch.no_grad():
            d_ref, m_ref = ref(lat, 500, encoder_hidden_states=text, controlnet_cond=cond, conditioning_mask=mask,
                               conditioning_scale=0.7, guess_mode=False, return_dict=False)
    finally:
        torch.Tensor.to = real_to
    with torch.no_grad():
        d, m = U.controlnet_forward(sd, cfg, lat.shape, 500, text, cond, mask, 0.7)
    assert len(d) == 12 and max((a - b).abs().max().item() for a, b in zip(d_ref, d)) < 1e-4
    assert (m_ref - m).abs().max().item() < 1e-4


def test_engine_pixel_condition_embedding_and_drop_in_class(backend):
    from motionclone_amd.models.sparse_controlnet import SparseControlNetModel
    dev = backend
    cfg = dict(U.TINY_CONFIG)
    csd = {k: v.half().float() for k, v in U.random_controlnet_state_dict(cfg, conditioning_channels=3,
                                                                            simplified=False).items()}
    F, H, W = 2, 8, 8
    text = torch.randn(2, 7, cfg["cross_attention_dim"], generator=torch.Generator().manual_seed(7)).half()
    cond, mask = pixel_cond_inputs(F, 8 * H, 8 * W)
    ceng = ControlNetEngine(csd, cfg, dev)

    def tok(t):
        return t.permute(0, 2, 3, 4, 1).reshape(-1, t.shape[1])
    # the embedding on its own (+ conv_in.bias, which the engine folds in)
    e = ceng._cond_embedding(cond.half().to(dev), mask.half().to(dev), F, H, W)
    with torch.no_grad():
        e_ref = U.cond_embedding_pyramid(csd, torch.cat([cond, mask], 1)) + csd["conv_in.bias"].reshape(1, -1, 1, 1, 1)
    assert rel(e, tok(e_ref)) < 5e-3
    # whole encoder
    down, mid = ceng.forward((2, 4, F, H, W), 500, text.to(dev), cond.half().to(dev), mask.half().to(dev), 0.8)
    with torch.no_grad():
        d_ref, m_ref = U.controlnet_forward(csd, cfg, (2, 4, F, H, W), 500, text.float(), cond, mask, 0.8)
    for a, b in zip(down, d_ref):
        assert rel(a, tok(b)) < 2e-2
    assert rel(mid, tok(m_ref)) < 2e-2
    with pytest.raises(ValueError, match="does not reduce to the sample grid"):
        ceng.forward((2, 4, F, H, W), 500, text.to(dev), cond[..., :32, :32].half().to(dev), mask[..., :32, :32].half().to(dev))
    # the drop-in class in the image_condition.yaml configuration: same keys, same outputs
    model = SparseControlNetModel(in_channels=4, block_out_channels=cfg["block_out_channels"], layers_per_block=2,
                                  cross_attention_dim=cfg["cross_attention_dim"], attention_head_dim=cfg["attention_heads"],
                                  set_noisy_sample_input_to_zero=True, use_simplified_condition_embedding=False,
                                  conditioning_channels=3,
                                  motion_module_kwargs=dict(num_attention_heads=cfg["motion_heads"],
                                                            attention_block_types=["Temporal_Self"],
                                                            temporal_position_encoding_max_len=32))
    assert not model.use_simplified_condition_embedding
    assert set(model.state_dict().keys()) == set(csd.keys())
    model.load_state_dict(csd)
    model = model.to(dev).half()
    d2, m2 = model(torch.zeros(2, 4, F, H, W, device=dev, dtype=torch.float16), 500, text.to(dev),
                   cond.half().to(dev), conditioning_mask=mask.half().to(dev), conditioning_scale=0.8, return_dict=False)
    assert all(torch.equal(a, b) for a, b in zip(d2, down)) and torch.equal(m2, mid)


def test_api_functions_with_the_pixel_condition_controlnet(backend):
    """obtain_motion_representation / sample_video with controlnet.use_simplified_condition_embedding == False
    (motionclone_functions.py:50-52,127-128): the condition is the preprocessed frame / the resized image itself."""
    from motionclone_amd.models.sparse_controlnet import SparseControlNetModel
    from test_dropin_api import build_pipeline
    dev = backend
    cfg = dict(U.TINY_CONFIG)
    sd = {k: v.half().float() for k, v in U.random_state_dict(cfg, seed=1234).items()}
    csd = {k: v.half().float() for k, v in U.random_controlnet_state_dict(cfg, conditioning_channels=3,
                                                                            simplified=False).items()}
    N, Gs, gscale, F = 3, 2, 0.3, 2
    pipe = build_pipeline(dev, cfg, sd, N, Gs, gscale)
    cn = SparseControlNetModel(in_channels=4, block_out_channels=cfg["block_out_channels"], layers_per_block=2,
                               cross_attention_dim=cfg["cross_attention_dim"], attention_head_dim=cfg["attention_heads"],
                               set_noisy_sample_input_to_zero=True, use_simplified_condition_embedding=False,
                               conditioning_channels=3,
                               motion_module_kwargs=dict(num_attention_heads=cfg["motion_heads"],
                                                         attention_block_types=["Temporal_Self"]))
    cn.load_state_dict(csd)
    pipe.controlnet = cn.to(dev).half()
    c = pipe.input_config
    c.video_length, c.image_index, c.controlnet_scale = F, [0], 0.8
    text = torch.randn(2, 7, cfg["cross_attention_dim"], generator=torch.Generator().manual_seed(7)).half()
    vid = (0.18215 * torch.randn(1, 4, F, 8, 8, generator=torch.Generator().manual_seed(11))).half()
    frames = (torch.rand(F, 3, 64, 64, generator=torch.Generator().manual_seed(12)) * 2 - 1).half()   # video_preprocess range
    rep = pipe.obtain_motion_representation(generator=torch.Generator(device=dev).manual_seed(5), use_controlnet=True,
                                            video_latents=vid.to(dev), uncond_embeddings=text[0:1].to(dev),
                                            video_data=frames.to(dev))
    noise = torch.randn(vid.shape, generator=torch.Generator(device=dev).manual_seed(5), device=dev, dtype=torch.float16).cpu()
    noisy = G.add_noise(G.alphas_cumprod(), 400, vid.float(), noise.float())
    cond = torch.zeros(1, 3, F, 64, 64)
    mask = torch.zeros(1, 1, F, 64, 64)
    cond[:, :, 0] = ((frames.float() + 1) / 2)[0]
    mask[:, :, 0] = 1
    with torch.no_grad():
        dr, mr = U.controlnet_forward(csd, cfg, noisy.shape, 400, text[[0]].float(), cond.half().float(), mask, 0.8)
        rec = {}
        U.unet_forward(sd, cfg, noisy, 400, text[[0]].float(), only_motion_feature=True, record=rec, down_residuals=dr,
                       mid_residual=mr)
        rep_ref = G.motion_representation(G.temp_attn_prob(rec, cfg["motion_heads"]))
    assert list(rep) == list(rep_ref)
    for k in rep_ref:
        assert (rep[k][0].float().cpu() - rep_ref[k][0]).abs().max() < 5e-3
    with pytest.raises(ValueError, match="needs the preprocessed frames"):
        pipe.obtain_motion_representation(use_controlnet=True, video_latents=vid.to(dev), uncond_embeddings=text[0:1].to(dev))

    # sampling: the condition image stays in pixel space (no VAE encode)
    image = torch.rand(1, 3, 1, 64, 64, generator=torch.Generator().manual_seed(13)).half()
    lat0 = torch.randn(1, 4, F, 8, 8, generator=torch.Generator().manual_seed(2025)).half()
    pipe.motion_representation_dict = rep
    out = pipe.sample_video(noisy_latents=lat0.to(dev), text_embeddings=text.to(dev), decode=False, add_controlnet=True,
                            controlnet_images=image.to(dev))
    cond2 = torch.zeros(1, 3, F, 64, 64)
    cond2[:, :, 0] = image[:, :, 0].float()
    ts = G.uneven_timesteps(N, Gs, gscale)
    hp = dict(cfg_scale=7.5, motion_guidance_weight=2000.0, warm_up_steps=10, cool_up_steps=10, guidance_steps=Gs)
    rep_cpu = {k: [a.float().cpu(), b.cpu()] for k, (a, b) in rep.items()}
    x = lat0.float()
    for s in range(N):
        with torch.no_grad():
            d, m = U.controlnet_forward(csd, cfg, (2, 4, F, 8, 8), int(ts[s]), text.float(), cond2, mask, 0.8)
        if s < Gs:
            x, _ = G.guided_step(sd, cfg, x, s, ts, text.float(), rep_cpu, hp, res_u=([t[[0]] for t in d], m[[0]]),
                                 res_c=([t[[1]] for t in d], m[[1]]))
        else:
            x, _ = G.plain_step_full(sd, cfg, x, s, ts, text.float(), 7.5, res=(d, m))
    assert rel(out, x) < 3e-2
