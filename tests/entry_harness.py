"""TEST INFRASTRUCTURE.  Executes the UNMODIFIED reference entry scripts (/root/reference/t2v_video_sample.py,
i2v_video_sample.py) `main(args)` against the drop-in `motionclone/` package of this repo, in a child process:

    python tests/entry_harness.py {t2v|i2v} WORKDIR

The third-party modules the scripts import but this image lacks are stood in for, the way a deployment would resolve
them: `diffusers.AutoencoderKL` / `transformers.CLIPTextModel` -> the in-tree HIP drop-ins,
`diffusers.DDIMScheduler` -> motionclone_amd.scheduler.DDIMSchedulerState, `omegaconf` -> a small yaml-backed
attribute dict, `decord` -> a reader of WORKDIR/*.mp4.npy, `imageio.mimwrite` -> np.save.  The scripts hard-code
device "cuda"; without a GPU the kernels run on the host simulator (tests/hipemu) and "cuda" is mapped to "cpu".
Tiny checkpoints, configs, a 4-frame "video" and condition images are written to WORKDIR; what entered and left the
UNet loop is recorded to WORKDIR/record.pt for the parent test to check against the oracle."""
import argparse
import importlib.machinery
import json
import os
import runpy
import sys
import types

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE_ROOT = os.environ.get("MC_REFERENCE_ROOT", "/root/reference")
F, PX, STEPS, GUIDED, GSCALE = 4, 16, 3, 2, 0.3


def px_of(kind):
    """frame size in pixels: the tiny VAE of the t2v / i2v runs has 2 levels (16 px -> 8 x 8 latents); the sketch ControlNet's
    condition pyramid always reduces by 8 (sparse_controlnet.py:124), so that run uses a 4-level VAE and 64 px frames"""
    return 64 if kind == "i2v_sketch" else PX


def vae_config_of(kind):
    from oracle import vae_ref as V
    cfg = dict(V.TINY_VAE_CONFIG)
    if kind == "i2v_sketch":
        cfg["block_out_channels"] = (64, 64, 64, 64)
    return cfg


# ---- stand-ins -------------------------------------------------------------------------------------------------------
class DictConfig(dict):
    """attribute-style dict (the slice of omegaconf.DictConfig the scripts use: attribute get / set, .get)"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return DictConfig({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _plain(x):
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


class OmegaConf:
    @staticmethod
    def load(path):
        with open(path) as f:
            return _wrap(yaml.safe_load(f) or {})

    @staticmethod
    def save(config, path):
        with open(path, "w") as f:
            yaml.safe_dump(_plain(config), f)

    @staticmethod
    def to_container(cfg, **kw):
        return _plain(cfg)


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


class _VideoReader:
    def __init__(self, path, *a, **k):
        self.frames = np.load(path + ".npy")

    def __len__(self):
        return len(self.frames)

    def get_avg_fps(self):
        return 8.0

    def get_batch(self, idx):
        return self.frames[np.asarray(idx)]


def install_stubs(work):
    import transformers
    from motionclone_amd.models.clip import CLIPTextModel
    from motionclone_amd.models.vae import AutoencoderKL
    from motionclone_amd.scheduler import DDIMSchedulerState
    transformers.CLIPTextModel = CLIPTextModel
    _module("omegaconf", OmegaConf=OmegaConf)
    _module("diffusers", AutoencoderKL=AutoencoderKL, DDIMScheduler=DDIMSchedulerState)
    _module("diffusers.utils")
    _module("diffusers.utils.import_utils", is_xformers_available=lambda: False)
    _module("decord", VideoReader=_VideoReader, bridge=types.SimpleNamespace(set_bridge=lambda *a, **k: None))
    written = []

    def mimwrite(path, frames, fps=8, **k):
        np.save(path + ".npy", np.stack([np.asarray(f) for f in frames]))
        written.append(path)
    _module("imageio", mimwrite=mimwrite, mimsave=mimwrite)
    return written


def map_cuda_to_cpu():
    """no GPU here: the kernels run on the host simulator, so tensors stay on the CPU"""
    def fix(a):
        if isinstance(a, str) and a.startswith("cuda"):
            return "cpu"
        if isinstance(a, torch.device) and a.type == "cuda":
            return torch.device("cpu")
        return a
    t_to, m_to = torch.Tensor.to, torch.nn.Module.to
    torch.Tensor.to = lambda self, *a, **k: t_to(self, *[fix(x) for x in a], **{n: fix(v) for n, v in k.items()})
    torch.nn.Module.to = lambda self, *a, **k: m_to(self, *[fix(x) for x in a], **{n: fix(v) for n, v in k.items()})


# ---- assets ----------------------------------------------------------------------------------------------------------
def tiny_clip_config():
    return dict(vocab_size=514, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5)


def write_tokenizer(path):
    os.makedirs(path, exist_ok=True)
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    chars = [chr(c) for c in cs]
    vocab = {}
    for ch in chars:
        vocab[ch] = len(vocab)
    for ch in chars:
        vocab[ch + "</w>"] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    json.dump(vocab, open(os.path.join(path, "vocab.json"), "w"))
    open(os.path.join(path, "merges.txt"), "w").write("#version: 0.2\n")
    json.dump({"model_max_length": 77, "tokenizer_class": "CLIPTokenizer"}, open(os.path.join(path, "tokenizer_config.json"), "w"))


# configs/t2v_camera.jsonl:1-8 of the reference = BASELINE config 3's eight (prompt, reference-video) pairs, one per GPU:
# (reference video, new_prompt, seed or None = the script's --default-seed).  Three distinct videos, four distinct seeds.
CAMERA8 = [("camera_zoom_in", "Relics on the seabed", 42), ("camera_zoom_in", "A road in the mountain", 42),
           ("camera_zoom_in", "Caves, a path for exploration", 2026), ("camera_zoom_in", "Railway for train", None),
           ("camera_zoom_out", "Tree, in the mountain", 2026), ("camera_zoom_out", "Red car on the track", 2026),
           ("camera_zoom_out", "Man, standing in his garden.", 2026), ("camera_1", "A island, on the ocean, sunny day", None)]


def write_assets(work, kind, n_examples=1, camera8=False):
    from motionclone_amd.models.clip import clip_param_shapes
    from oracle import unet3d_ref as U
    from oracle import vae_ref as V
    cfg = dict(U.TINY_CONFIG)
    sd = {k: v.half().float() for k, v in U.random_state_dict(cfg, seed=1234).items()}
    root = os.path.join(work, "sd")
    for sub in ("unet", "vae", "text_encoder"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    # SD "unet/": config.json + the 2-D weights; the motion modules arrive through load_weights(motion_module_path=...)
    json.dump(dict(sample_size=8, in_channels=4, out_channels=4, block_out_channels=list(cfg["block_out_channels"]),
                   layers_per_block=2, cross_attention_dim=cfg["cross_attention_dim"],
                   attention_head_dim=cfg["attention_heads"], norm_num_groups=32, norm_eps=1e-5, act_fn="silu",
                   _class_name="UNet2DConditionModel"), open(os.path.join(root, "unet", "config.json"), "w"))
    torch.save({k: v for k, v in sd.items() if "motion_modules." not in k},
               os.path.join(root, "unet", "diffusion_pytorch_model.bin"))
    torch.save({"state_dict": {k: v for k, v in sd.items() if "motion_modules." in k}}, os.path.join(work, "mm.ckpt"))
    vcfg = vae_config_of(kind)
    px = px_of(kind)
    vsd = {k: v.half().float() for k, v in V.random_state_dict(vcfg, seed=77).items()}
    json.dump(dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=list(vcfg["block_out_channels"]),
                   layers_per_block=vcfg["layers_per_block"], norm_num_groups=vcfg["norm_num_groups"],
                   scaling_factor=0.18215, sample_size=px, act_fn="silu", _class_name="AutoencoderKL"),
              open(os.path.join(root, "vae", "config.json"), "w"))
    torch.save(vsd, os.path.join(root, "vae", "diffusion_pytorch_model.bin"))
    ccfg = tiny_clip_config()
    g = torch.Generator().manual_seed(5)
    csd = {}
    for name, shape in clip_param_shapes(ccfg).items():
        if "layer_norm" in name:
            csd[name] = (torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)) + 0.05 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            csd[name] = 0.02 * torch.randn(shape, generator=g)
        else:
            csd[name] = torch.randn(shape, generator=g) * (0.5 if "embedding" in name else shape[-1] ** -0.5)
    csd = {k: v.half().float() for k, v in csd.items()}
    json.dump(ccfg, open(os.path.join(root, "text_encoder", "config.json"), "w"))
    torch.save(csd, os.path.join(root, "text_encoder", "pytorch_model.bin"))
    write_tokenizer(os.path.join(root, "tokenizer"))
    # configs (same keys as configs/t2v_camera.yaml, configs/model_config/model_config.yaml, configs/i2v_rgb.yaml)
    mk = dict(num_attention_heads=cfg["motion_heads"], num_transformer_block=1,
              attention_block_types=["Temporal_Self", "Temporal_Self"], temporal_position_encoding=True,
              temporal_position_encoding_max_len=32, temporal_attention_dim_div=1, zero_initialize=True)
    model_config = dict(unet_additional_kwargs=dict(use_inflated_groupnorm=True, use_motion_module=True,
                                                    motion_module_resolutions=[1, 2, 4, 8], motion_module_mid_block=False,
                                                    motion_module_type="Vanilla", motion_module_kwargs=mk),
                        noise_scheduler_kwargs=dict(beta_start=0.00085, beta_end=0.012, beta_schedule="linear",
                                                    steps_offset=1, clip_sample=False))
    yaml.safe_dump(model_config, open(os.path.join(work, "model_config.yaml"), "w"))
    infer = dict(motion_module=os.path.join(work, "mm.ckpt"), dreambooth_path="", model_config=os.path.join(work, "model_config.yaml"),
                 cfg_scale=7.5, negative_prompt="bad quality", postive_prompt=" 8k", inference_steps=STEPS, guidance_scale=GSCALE,
                 guidance_steps=GUIDED, warm_up_steps=10, cool_up_steps=10, motion_guidance_weight=2000,
                 motion_guidance_blocks=["up_blocks.1"], add_noise_step=400)
    rng = np.random.RandomState(3)
    np.save(os.path.join(work, "clip.mp4.npy"), rng.randint(0, 256, size=(9, 20, 24, 3)).astype(np.uint8))
    example = dict(video_path=os.path.join(work, "clip.mp4"), new_prompt="a cat runs", seed=42)
    if kind in ("i2v", "i2v_sketch"):
        from PIL import Image
        mk1 = dict(mk, attention_block_types=["Temporal_Self"])
        simplified = kind == "i2v"     # configs/sparsectrl/latent_condition.yaml vs image_condition.yaml
        yaml.safe_dump(dict(controlnet_additional_kwargs=dict(
            set_noisy_sample_input_to_zero=True, use_simplified_condition_embedding=simplified,
            conditioning_channels=4 if simplified else 3,
            use_motion_module=True, motion_module_resolutions=[1, 2, 4, 8], motion_module_mid_block=False,
            motion_module_type="Vanilla", motion_module_kwargs=mk1)), open(os.path.join(work, "cn_config.yaml"), "w"))
        cnsd = {k: v.half().float() for k, v in U.random_controlnet_state_dict(
            cfg, conditioning_channels=4 if simplified else 3, simplified=simplified).items()}
        torch.save({"controlnet": cnsd}, os.path.join(work, "v3_sd15_sparsectrl_rgb.ckpt"))
        infer.update(controlnet_path=os.path.join(work, "v3_sd15_sparsectrl_rgb.ckpt"),
                     controlnet_config=os.path.join(work, "cn_config.yaml"), adapter_lora_path="", guidance_steps=GUIDED)
        img = os.path.join(work, "cond0.png")
        Image.fromarray(rng.randint(0, 256, size=(px, px, 3)).astype(np.uint8)).save(img)
        example.update(condition_image_paths=[img], image_index=[0], controlnet_scale=0.8)
    yaml.safe_dump(infer, open(os.path.join(work, "infer.yaml"), "w"))
    if camera8:
        assert kind == "t2v"
        for n, stem in enumerate(sorted({c[0] for c in CAMERA8})):
            np.save(os.path.join(work, stem + ".mp4.npy"), np.random.RandomState(30 + n).randint(0, 256, size=(9, 20, 24, 3)).astype(np.uint8))
        with open(os.path.join(work, "examples.jsonl"), "w") as f:
            for stem, prompt, seed in CAMERA8:
                line = dict(video_path=os.path.join(work, stem + ".mp4"), new_prompt=prompt)
                if seed is not None:
                    line["seed"] = seed
                f.write(json.dumps(line) + "\n")
        return cfg
    with open(os.path.join(work, "examples.jsonl"), "w") as f:
        f.write(json.dumps(example) + "\n")
        for n in range(1, n_examples):     # same reference video (the .pt is overwritten per example, as in the reference)
            f.write(json.dumps(dict(example, new_prompt="a dog walks %d" % n, seed=2026 + n)) + "\n")
    return cfg


# ---- recording -------------------------------------------------------------------------------------------------------
def install_recorders(rec):
    from motionclone_amd.models.sparse_controlnet import SparseControlNetModel
    from motionclone_amd.models.unet import UNet3DConditionModel
    from motionclone_amd.sampler import MotionCloneSampler
    fwd = UNet3DConditionModel.forward

    def forward(self, sample, timestep, encoder_hidden_states, *a, **k):
        if k.get("only_motion_feature", a[5] if len(a) > 5 else False):
            rec["extract"] = dict(noisy=sample.detach().clone(), t=int(timestep), text=encoder_hidden_states.detach().clone())
        return fwd(self, sample, timestep, encoder_hidden_states, *a, **k)
    UNet3DConditionModel.forward = forward
    step = MotionCloneSampler.step

    def rstep(self, latents, i, text, rep_dev, aux=None, ctrl=None, **kw):
        if i == 0:
            rec["loop"] = dict(lat0=latents.detach().clone(), text=text.detach().clone(), timesteps=[int(t) for t in self.timesteps],
                               G=self.G, ctrl=None if ctrl is None else {k: (v.detach().clone() if torch.is_tensor(v) else v)
                                                                         for k, v in ctrl.items()})
        assert not kw.get("eta"), "the entry scripts sample with eta = 0"
        out = step(self, latents, i, text, rep_dev, aux=aux, ctrl=ctrl, **kw)
        rec["loop"]["last"] = out.detach().clone()
        rec["loop"]["steps_run"] = i + 1
        return out
    MotionCloneSampler.step = rstep
    cfw = SparseControlNetModel.forward

    def cforward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_mask=None,
                 conditioning_scale=1.0, **k):
        rec.setdefault("controlnet_calls", []).append(dict(t=int(timestep), B=sample.shape[0], scale=float(conditioning_scale),
                                                           cond=controlnet_cond.detach().clone(), mask=conditioning_mask.detach().clone()))
        return cfw(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_mask, conditioning_scale, **k)
    SparseControlNetModel.forward = cforward


def main():
    """entry_harness.py KIND WORKDIR [--examples N | --camera8] [--launch [--lanes K]]   (--camera8: the eight lines of the
    reference's configs/t2v_camera.jsonl as the examples file; --launch: through motionclone_amd.launch, sharded
    over the torchrun environment's ranks and K lanes per rank; the assets must already exist in WORKDIR, outputs go to
    WORKDIR/videos_rank<r>)"""
    kind, work = sys.argv[1], os.path.abspath(sys.argv[2])
    n_examples = int(sys.argv[sys.argv.index("--examples") + 1]) if "--examples" in sys.argv else 1
    launch = "--launch" in sys.argv
    n_lanes = sys.argv[sys.argv.index("--lanes") + 1] if "--lanes" in sys.argv else "1"
    os.makedirs(work, exist_ok=True)
    sys.path.insert(0, ROOT)        # `motionclone` must resolve to this repo's drop-in package, not to the reference's
    assert not any(os.path.abspath(p) == REFERENCE_ROOT for p in sys.path)
    if not torch.cuda.is_available():
        from motionclone_amd import build, lib
        lib.use_library_for_tests(build.build_emu())
        map_cuda_to_cpu()
    written = install_stubs(work)
    if not launch:
        write_assets(work, kind, n_examples, camera8="--camera8" in sys.argv)
    rec = {}
    install_recorders(rec)
    script = os.path.join(REFERENCE_ROOT, "t2v_video_sample.py" if kind == "t2v" else "i2v_video_sample.py")
    px = px_of(kind)
    common = dict(pretrained_model_path=os.path.join(work, "sd"), inference_config=os.path.join(work, "infer.yaml"),
                  examples=os.path.join(work, "examples.jsonl"))
    if launch:
        from motionclone_amd import launch as L
        rank = int(os.environ.get("RANK", 0))
        L.main([script, "--pretrained-model-path", common["pretrained_model_path"], "--inference_config",
                common["inference_config"], "--examples", common["examples"], "--motion-representation-save-dir",
                os.path.join(work, "mr_sharded"), "--generated-videos-save-dir", os.path.join(work, "videos_rank%d" % rank),
                "--L", str(F), "--W", str(px), "--H", str(px), "--vae-scale", "8" if kind == "i2v_sketch" else "2", "--lanes", n_lanes])
        print("ENTRY_OK", kind, written)
        return
    ns = runpy.run_path(script, run_name="entry_script_under_test")
    import motionclone.models.unet as mu
    assert os.path.abspath(mu.__file__).startswith(ROOT), mu.__file__
    args = argparse.Namespace(motion_representation_save_dir=os.path.join(work, "motion_representation"),
                              generated_videos_save_dir=os.path.join(work, "generated_videos"), visible_gpu=None,
                              default_seed=2025, L=F, W=px, H=px, without_xformers=False, **common)
    ns["main"](args)
    rec["videos"] = written
    torch.save(rec, os.path.join(work, "record.pt"))
    print("ENTRY_OK", kind, written)


if __name__ == "__main__":
    main()
