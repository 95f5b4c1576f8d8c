"""ORACLE - TEST INFRASTRUCTURE ONLY (authoring container only; /root/reference does not travel).

Makes the reference's own, unmodified Python modules importable where diffusers / xformers /
decord / omegaconf / torchvision / cv2 / imageio are not installed, by registering minimal stand-ins
for the handful of third-party symbols the hot path touches.  The stand-ins
restate the *published behaviour* of diffusers==0.16.0 (pinned at environment.yaml:13); none of the
reference's source is copied.  Used by tests/test_oracle_pins.py and tests/golden/make_golden.py.
"""
import importlib
import importlib.machinery
import inspect
import math
import os
import sys
import types

import torch
from torch import nn

REFERENCE_ROOT = os.environ.get("MC_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "motionclone"))


class _AutoModule(types.ModuleType):
    """A module whose unknown attributes resolve to inert placeholder classes."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


def _mod(name, **attrs):
    m = _AutoModule(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


class FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def register_to_config(init):
    """Record every __init__ argument (defaults included) as `self.config` (diffusers ConfigMixin)."""
    sig = inspect.signature(init)

    def wrapped(self, *args, **kwargs):
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        cfg.pop("kwargs", None)
        self._internal_dict = FrozenDict(cfg)
        init(self, *args, **kwargs)

    return wrapped


class ConfigMixin:
    @property
    def config(self):
        return self._internal_dict

    @classmethod
    def from_config(cls, config, **extra):
        params = inspect.signature(cls.__init__).parameters
        kw = {k: v for k, v in dict(config).items() if k in params}
        kw.update({k: v for k, v in extra.items() if k in params})
        return cls(**kw)


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def enable_xformers_memory_efficient_attention(self, op=None):
        for m in self.modules():
            if m is not self and hasattr(m, "set_use_memory_efficient_attention_xformers"):
                m.set_use_memory_efficient_attention_xformers(True, op)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * torch.nn.functional.gelu(gate)


class FeedForward(nn.Module):
    """diffusers FeedForward(dim, activation_fn='geglu'): [GEGLU(dim, 4 dim), Dropout, Linear(4 dim, dim)]."""

    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False):
        super().__init__()
        assert activation_fn == "geglu"
        inner = dim * mult
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out or dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.n, self.flip, self.shift = num_channels, flip_sin_to_cos, downscale_freq_shift

    def forward(self, t):
        half = self.n // 2
        freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32, device=t.device)
                          / (half - self.shift))
        e = t[:, None].float() * freqs[None]
        emb = torch.cat([torch.sin(e), torch.cos(e)], dim=-1)
        if self.flip:
            emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
        return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu"):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, x):
        return self.linear_2(self.act(self.linear_1(x)))


class BaseOutput(dict):
    """dataclass/dict hybrid; the path only reads `.sample`."""

    def __post_init__(self):
        for k, v in self.__dict__.items():
            self[k] = v


class DDIMScheduler:
    """The slice of diffusers DDIMScheduler state the path consumes (restated, see oracle/guidance_ref.py)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 steps_offset=0, clip_sample=True, set_alpha_to_one=True, prediction_type="epsilon", **kw):
        assert beta_schedule == "linear"
        betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.config = FrozenDict(num_train_timesteps=num_train_timesteps, steps_offset=steps_offset,
                                 clip_sample=clip_sample, prediction_type=prediction_type, thresholding=False,
                                 clip_sample_range=1.0)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _get_variance(self, t, t_prev):
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[t_prev] if t_prev >= 0 else self.final_alpha_cumprod
        return ((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)

    def step(self, model_output, timestep, sample, eta=0.0, generator=None):
        raise NotImplementedError


def _logging_module():
    import logging as pylog

    class _L(types.ModuleType):
        @staticmethod
        def get_logger(name=None):
            return pylog.getLogger(name or "diffusers")
    m = _L("diffusers.utils.logging")
    return m


_installed = False


def install():
    """Register the stand-ins and put the reference on sys.path.  Idempotent."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    import transformers  # noqa: F401  (real; must be imported before the fake torchvision appears)
    try:
        from transformers import CLIPTextModel, CLIPTokenizer  # noqa: F401
    except Exception:
        pass
    logging_mod = _logging_module()
    _mod("diffusers", __version__="0.16.0", DDIMScheduler=DDIMScheduler)
    _mod("diffusers.utils", BaseOutput=BaseOutput, logging=logging_mod, WEIGHTS_NAME="diffusion_pytorch_model.bin",
         deprecate=lambda *a, **k: None, is_accelerate_available=lambda: False)
    sys.modules["diffusers.utils.logging"] = logging_mod
    _mod("diffusers.utils.import_utils", is_xformers_available=lambda: False, BACKENDS_MAPPING={})
    _mod("diffusers.utils.torch_utils",
         randn_tensor=lambda shape, generator=None, device=None, dtype=None, layout=None:
         torch.randn(shape, generator=generator, device=device, dtype=dtype))
    _mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config,
         FrozenDict=FrozenDict)
    _mod("diffusers.models")
    _mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    _mod("diffusers.models.attention", FeedForward=FeedForward)
    _mod("diffusers.models.embeddings", Timesteps=Timesteps, TimestepEmbedding=TimestepEmbedding)
    _mod("diffusers.models.unet_2d_condition")
    _mod("diffusers.pipeline_utils")
    _mod("diffusers.pipelines")
    _mod("diffusers.pipelines.stable_diffusion")
    _mod("diffusers.pipelines.stable_diffusion.safety_checker")
    _mod("diffusers.pipelines.paint_by_example")
    _mod("diffusers.schedulers", DDIMScheduler=DDIMScheduler)
    for name in ("torchvision", "torchvision.transforms", "xformers", "xformers.ops", "decord", "cv2", "imageio",
                 "omegaconf"):
        if name not in sys.modules:
            _mod(name)
    sys.modules["decord"].bridge = types.SimpleNamespace(set_bridge=lambda *a, **k: None)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


SD15_UNET_CONFIG = dict(  # the unet/config.json values of SD-1.5 the path depends on
    sample_size=64, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
    block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1, mid_block_scale_factor=1,
    act_fn="silu", norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=768, attention_head_dim=8,
)

MOTION_KWARGS = dict(  # configs/model_config/model_config.yaml:1-14
    use_inflated_groupnorm=True, use_motion_module=True, motion_module_resolutions=[1, 2, 4, 8],
    motion_module_mid_block=False, motion_module_type="Vanilla",
    motion_module_kwargs=dict(num_attention_heads=8, num_transformer_block=1,
                              attention_block_types=["Temporal_Self", "Temporal_Self"],
                              temporal_position_encoding=True, temporal_attention_dim_div=1, zero_initialize=True),
)


def reference_unet(oracle_cfg):
    """Instantiate the reference's UNet3DConditionModel for an oracle config dict (oracle/unet3d_ref.py)."""
    install()
    from motionclone.models.unet import UNet3DConditionModel
    kw = dict(SD15_UNET_CONFIG)
    kw.update(block_out_channels=tuple(oracle_cfg["block_out_channels"]),
              cross_attention_dim=oracle_cfg["cross_attention_dim"], attention_head_dim=oracle_cfg["attention_heads"],
              layers_per_block=oracle_cfg["layers_per_block"])
    mk = dict(MOTION_KWARGS)
    mk["motion_module_kwargs"] = dict(MOTION_KWARGS["motion_module_kwargs"],
                                      num_attention_heads=oracle_cfg["motion_heads"])
    kw.update(mk)
    kw.update(unet_use_cross_frame_attention=False, unet_use_temporal_attention=False)
    return UNet3DConditionModel(**kw)


class RefHarness:
    """The reference's guidance functions bound the way t2v_video_sample.py:57-72 binds them, on a
    reference UNet carrying a given state-dict; VAE / CLIP / file I/O are bypassed with tensors."""

    def __init__(self, oracle_cfg, sd, hp, num_inference_steps, guidance_steps, guidance_scale):
        install()
        import motionclone.utils.motionclone_functions as mf
        self.mf = mf
        unet = reference_unet(oracle_cfg)
        missing, unexpected = unet.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        assert all("pos_encoder" in m for m in missing), missing
        unet.eval()
        for p in unet.parameters():
            p.requires_grad = False
        cfgns = types.SimpleNamespace(motion_guidance_blocks=["up_blocks.1"], guidance_steps=guidance_steps,
                                      warm_up_steps=hp["warm_up_steps"], cool_up_steps=hp["cool_up_steps"],
                                      cfg_scale=hp["cfg_scale"], motion_guidance_weight=hp["motion_guidance_weight"])
        sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1,
                              clip_sample=False)
        sched.customized_step = mf.schedule_customized_step.__get__(sched)
        sched.customized_set_timesteps = mf.schedule_set_timesteps.__get__(sched)
        unet.forward = mf.unet_customized_forward.__get__(unet)
        pipe = types.SimpleNamespace(unet=unet, scheduler=sched, input_config=cfgns, add_controlnet=False)
        unet.input_config = cfgns
        for fn in ("single_step_video", "get_temp_attn_prob", "compute_temp_loss", "add_noise"):
            setattr(pipe, fn, getattr(mf, fn).__get__(pipe))
        mf.prep_unet_attention(unet, cfgns.motion_guidance_blocks)
        mf.prep_unet_conv(unet)
        sched.customized_set_timesteps(num_inference_steps, guidance_steps, guidance_scale, device="cpu",
                                       timestep_spacing_type="uneven")
        self.pipe, self.unet, self.sched = pipe, unet, sched

    def extract(self, video_latents, noise, uncond_text, add_noise_step=400):
        """obtain_motion_representation (motionclone_functions.py:40-43,74-79) with VAE/CLIP bypassed."""
        with torch.no_grad():
            noisy = self.pipe.add_noise(add_noise_step, video_latents, noise)
            self.unet(noisy, add_noise_step, encoder_hidden_states=uncond_text, return_dict=False,
                      only_motion_feature=True)
            prob = self.pipe.get_temp_attn_prob()
            return {k: [v, i.to(torch.uint8)] for k, t in prob.items() for v, i in [torch.topk(t, k=1, dim=-1)]}

    def step(self, latents, step_index, text, rep):
        self.pipe.text_embeddings = text
        self.pipe.motion_representation_dict = rep
        self.pipe.motion_scale = self.pipe.input_config.motion_guidance_weight
        return self.pipe.single_step_video(latents, step_index, self.sched.timesteps[step_index], {})
