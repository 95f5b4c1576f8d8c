"""VAE decode: HIP decoder engine vs the oracle's restatement of diffusers 0.16.0 AutoencoderKL
(parity unpinned: no diffusers in the image), the drop-in AutoencoderKL surface, and decode_latents through the pipeline."""
import numpy as np
import pytest
import torch

from motionclone_amd import ops
from motionclone_amd.vae_engine import VaeDecoderEngine
from oracle import vae_ref as V


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


@pytest.fixture
def tiny_vae():
    cfg = dict(V.TINY_VAE_CONFIG)
    sd = {k: v.half().float() for k, v in V.random_state_dict(cfg, seed=77).items()}
    return cfg, sd


def test_softmax_rows_and_video_post(backend):
    dev = backend
    x = (torch.randn(70, 128, generator=torch.Generator().manual_seed(1)) * 4).half().to(dev)
    ref = torch.softmax(x.float(), -1)
    got = ops.softmax_rows_(x.clone())
    assert (got.float() - ref).abs().max() < 2e-3
    assert (got.float().sum(-1) - 1).abs().max() < 5e-3
    t = (torch.randn(2 * 6 * 5, 4, generator=torch.Generator().manual_seed(2)) * 1.5).half().to(dev)
    v = ops.video_post(t, 3, 2, 6, 5)
    want = (t.float().reshape(2, 30, 4)[:, :, :3].permute(2, 0, 1).reshape(1, 3, 2, 6, 5) / 2 + 0.5).clamp(0, 1)
    assert v.dtype == torch.float32 and torch.allclose(v, want, atol=1e-6)


def test_decoder_matches_oracle(backend, tiny_vae):
    dev = backend
    cfg, sd = tiny_vae
    eng = VaeDecoderEngine(sd, cfg, dev)
    z = torch.randn(3, 4, 8, 8, generator=torch.Generator().manual_seed(5)).half()
    got = eng.decode(z.to(dev))
    with torch.no_grad():
        ref = V.decode(sd, cfg, z.float())
    assert got.shape == ref.shape == (3, 3, 16, 16)
    assert rel(got, ref) < 2e-2, rel(got, ref)
    # frames are independent: a chunked decode equals the batched one
    one = eng.decode(z[1:2].to(dev))
    assert rel(one, got[1:2]) < 2e-3


def test_decode_video_matches_oracle(backend, tiny_vae):
    dev = backend
    cfg, sd = tiny_vae
    eng = VaeDecoderEngine(sd, cfg, dev)
    lat = (0.18215 * torch.randn(1, 4, 3, 8, 8, generator=torch.Generator().manual_seed(9))).half()
    got = eng.decode_video(lat.to(dev))
    with torch.no_grad():
        ref = V.decode_latents(sd, cfg, lat.float())
    assert got.dtype == torch.float32 and got.shape == ref.shape == (1, 3, 3, 16, 16)
    assert float(got.min()) >= 0.0 and float(got.max()) <= 1.0
    assert (got.cpu() - ref).abs().max() < 2e-2


def test_autoencoderkl_dropin_and_pipeline(backend, tiny_vae):
    dev = backend
    cfg, sd = tiny_vae
    from motionclone.models.vae import AutoencoderKL
    from motionclone_amd.pipelines.pipeline_animation import AnimationPipeline
    vae = AutoencoderKL(block_out_channels=cfg["block_out_channels"], layers_per_block=cfg["layers_per_block"])
    assert set(vae.state_dict().keys()) == set(V.param_shapes(cfg).keys())
    vae.load_state_dict({k: v.half() for k, v in sd.items()})
    vae = vae.to(dev)
    assert vae.config.scaling_factor == 0.18215 and vae.dtype == torch.float16
    z = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(5)).half().to(dev)
    with torch.no_grad():
        ref = V.decode(sd, cfg, z.float().cpu())
    assert rel(vae.decode(z).sample, ref) < 2e-2
    pipe = AnimationPipeline(vae=vae)
    assert pipe.vae_scale_factor == 2
    lat = (0.18215 * torch.randn(1, 4, 2, 8, 8, generator=torch.Generator().manual_seed(3))).half().to(dev)
    video = pipe.decode_latents(lat)
    with torch.no_grad():
        want = V.decode_latents(sd, cfg, lat.float().cpu()).numpy()
    assert isinstance(video, np.ndarray) and video.dtype == np.float32 and video.shape == (1, 3, 2, 16, 16)
    assert np.abs(video - want).max() < 2e-2
    # encode side (motionclone_functions.py:64-65): latent_dist.sample() * scaling_factor, noise from torch's generator
    img = (torch.rand(2, 3, 16, 16, generator=torch.Generator().manual_seed(8)) * 2 - 1).half()
    dist = vae.encode(img.to(dev)).latent_dist
    gen = torch.Generator(device=dev).manual_seed(123)
    smp = dist.sample(gen)
    noise = torch.randn((2, 4, 8, 8), generator=torch.Generator(device=dev).manual_seed(123), device=dev, dtype=torch.float16)
    with torch.no_grad():
        want = V.encode_sample(sd, cfg, img.float(), noise.float().cpu())
        mean, _ = V.encode_moments(sd, cfg, img.float())
    assert smp.shape == (2, 4, 8, 8) and rel(smp, want) < 2e-2
    assert rel(dist.mode(), mean) < 2e-2


def test_encoder_matches_oracle(backend, tiny_vae):
    dev = backend
    cfg, sd = tiny_vae
    from motionclone_amd.vae_engine import VaeEncoderEngine
    eng = VaeEncoderEngine(sd, cfg, dev)
    img = (torch.rand(3, 3, 16, 16, generator=torch.Generator().manual_seed(4)) * 2 - 1).half()
    dist = eng.encode(img.to(dev))
    with torch.no_grad():
        mean, std = V.encode_moments(sd, cfg, img.float())
    assert rel(dist.mode(), mean) < 2e-2, rel(dist.mode(), mean)
    noise = torch.ones(3, 4, 8, 8, dtype=torch.float16)
    one = ops.vae_sample(dist._tok, noise.to(dev), 4)
    assert rel(one.float().cpu() - dist.mode().float().cpu(), std) < 3e-2     # mean + std * 1


@pytest.mark.gpu
def test_full_size_decoder_one_frame_vs_oracle():
    """SD-1.5 VAE architecture (83.7 M decoder parameters), 2 frames of 32x32 latents -> 256x256, against the fp32 oracle on
    the host cores; plus determinism and frame independence at the config-2 size (64x64 latents -> 512x512)."""
    from motionclone_amd import lib
    lib._lib = None
    lib._is_emulated = False
    lib.load()
    dev = torch.device("cuda:0")
    cfg = dict(V.SD15_VAE_CONFIG)
    sd = {k: v.half().float() for k, v in V.random_state_dict(cfg, seed=4242).items() if not k.startswith("encoder") and not k.startswith("quant")}
    eng = VaeDecoderEngine(sd, cfg, dev)
    z = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(5)).half()
    got = eng.decode(z.to(dev))
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        ref = V.decode(sd, cfg, z.float())
    assert got.shape == (2, 3, 256, 256) and rel(got, ref) < 2e-2, rel(got, ref)
    lat = (0.18215 * torch.randn(1, 4, 16, 64, 64, generator=torch.Generator().manual_seed(11))).half().to(dev)
    v1 = eng.decode_video(lat)
    v2 = eng.decode_video(lat)
    assert v1.shape == (1, 3, 16, 512, 512) and torch.equal(v1, v2)
    assert torch.isfinite(v1).all() and float(v1.min()) >= 0 and float(v1.max()) <= 1
    single = eng.decode_video(lat[:, :, 5:6])
    assert (single[0, :, 0] - v1[0, :, 5]).abs().max() < 5e-3
    # encoder: 2 frames of 256x256 against the oracle
    from motionclone_amd.vae_engine import VaeEncoderEngine
    sde = {k: v.half().float() for k, v in V.random_state_dict(cfg, seed=4242).items() if k.startswith("encoder") or k.startswith("quant")}
    enc = VaeEncoderEngine(sde, cfg, dev)
    img = (torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(4)) * 2 - 1).half()
    dist = enc.encode(img.to(dev))
    with torch.no_grad():
        mean, std = V.encode_moments(sde, cfg, img.float())
    assert rel(dist.mode(), mean) < 2e-2, rel(dist.mode(), mean)
