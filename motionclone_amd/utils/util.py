"""Host-side helpers of the path (reference motionclone/utils/util.py): the names the entry scripts and the guidance
layer import from `motionclone.utils.util` - load_weights, auto_download, video_preprocess, classify_blocks,
set_all_seed, save_videos_grid, zero_rank_print.  Key maps / LoRA merges live in utils/convert.py."""
import os
import random

import numpy as np
import torch


def classify_blocks(block_list, name):
    """substring match of a module name against the configured guidance blocks (reference util.py:434-440)"""
    return any(block in name for block in block_list)


def set_all_seed(seed):
    """reference util.py:442-447 (inside a launcher lane the thread's private serial stream is seeded as well: lanes.py)"""
    from .. import lanes
    lanes.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


# checkpoints published under the two AnimateDiff hub repositories (reference util.py:45-81)
HUB_FILES = {
    "guoyww/animatediff": ("mm_sd_v14.ckpt", "mm_sd_v15.ckpt", "mm_sd_v15_v2.ckpt", "v3_sd15_mm.ckpt",
                           "v2_lora_PanLeft.ckpt", "v2_lora_PanRight.ckpt", "v2_lora_RollingAnticlockwise.ckpt",
                           "v2_lora_RollingClockwise.ckpt", "v2_lora_TiltDown.ckpt", "v2_lora_TiltUp.ckpt",
                           "v2_lora_ZoomIn.ckpt", "v2_lora_ZoomOut.ckpt", "v3_sd15_adapter.ckpt",
                           "v3_sd15_sparsectrl_rgb.ckpt", "v3_sd15_sparsectrl_scribble.ckpt"),
    "guoyww/animatediff_t2i_backups": ("realisticVisionV60B1_v51VAE.safetensors", "majicmixRealistic_v4.safetensors",
                                       "leosamsFilmgirlUltra_velvia20Lora.safetensors", "toonyou_beta3.safetensors",
                                       "majicmixRealistic_v5Preview.safetensors", "rcnzCartoon3d_v10.safetensors",
                                       "lyriel_v16.safetensors", "leosamsHelloworldXL_filmGrain20.safetensors",
                                       "TUSUN.safetensors"),
}


def auto_download(local_path, is_dreambooth_lora=False):
    """reference util.py:101-113: nothing to do when `local_path` exists; otherwise fetch that one file from the AnimateDiff
    hub repository into its folder (only names the repository is known to hold are accepted)."""
    if os.path.exists(local_path):
        return
    repo = "guoyww/animatediff_t2i_backups" if is_dreambooth_lora else "guoyww/animatediff"
    folder, filename = os.path.split(local_path)
    print(f"local file {local_path} does not exist. trying to download from {repo}")
    assert filename in HUB_FILES[repo], f"{filename} dose not exist in {repo}"
    folder = folder or "."
    os.makedirs(folder, exist_ok=True)
    from huggingface_hub import snapshot_download
    snapshot_download(repo_id=repo, local_dir=folder, allow_patterns=[filename])


def zero_rank_print(s):
    """print on rank 0 only (the reference's condition at util.py:83-84 can never be true; this is the intended one)"""
    import torch.distributed as dist
    if (not dist.is_available()) or (not dist.is_initialized()) or dist.get_rank() == 0:
        print("### " + s)


def save_videos_grid(videos, path, rescale=False, n_rows=6, fps=8):
    """reference util.py:87-99: [b, c, t, h, w] in [0, 1] (or [-1, 1] with rescale) -> one tiled clip (2-pixel padding like
    torchvision.utils.make_grid, which is not part of this image) written through imageio."""
    v = torch.as_tensor(videos).detach().float().cpu()
    b, c, t, h, w = v.shape
    cols = min(n_rows, b)
    rows = (b + cols - 1) // cols
    pad = 2 if b > 1 else 0
    frames = []
    for ti in range(t):
        grid = torch.zeros(c, rows * (h + pad) + pad, cols * (w + pad) + pad)
        for k in range(b):
            y, x = pad + (k // cols) * (h + pad), pad + (k % cols) * (w + pad)
            grid[:, y:y + h, x:x + w] = v[k, :, ti]
        if c == 1:
            grid = grid.expand(3, -1, -1)
        img = grid.permute(1, 2, 0)
        if rescale:
            img = (img + 1.0) / 2.0
        frames.append((img * 255).numpy().astype(np.uint8))
    if os.path.dirname(path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
    import imageio
    imageio.mimsave(path, frames, fps=fps)


def _load_checkpoint_file(path):
    from .. import checkpoints
    return checkpoints.read(path)


def _unwrap(sd):
    sd = sd["state_dict"] if "state_dict" in sd else sd
    sd.pop("animatediff_config", "")
    return sd


def load_weights(animation_pipeline, motion_module_path="", motion_module_lora_configs=(), adapter_lora_path="",
                 adapter_lora_scale=1.0, dreambooth_model_path="", lora_model_path="", lora_alpha=0.8):
    """reference util.py:115-215, same argument names and order of operations: motion module -> DreamBooth / base
    checkpoint in the original Stable-Diffusion layout (VAE, UNet, text encoder) -> kohya LoRA -> domain-adapter LoRA ->
    motion LoRAs.  The key maps and merges live in utils/convert.py; after any of them the packed HIP weights are
    rebuilt on next use."""
    from .convert import (convert_ldm_clip_checkpoint_concise, convert_ldm_unet_checkpoint, convert_ldm_vae_checkpoint,
                          convert_lora, load_diffusers_lora)
    pipeline = animation_pipeline
    if motion_module_path != "":
        print(f"load motion module from {motion_module_path}")
        sd = _unwrap(_load_checkpoint_file(motion_module_path))
        sd = {k: v for k, v in sd.items() if "motion_modules." in k}
        missing, unexpected = pipeline.unet.load_state_dict(sd, strict=False)
        if unexpected:   # tolerated like the reference (util.py:136-137 has the assert commented out)
            print(f"### motion module: {len(unexpected)} unexpected keys ignored, e.g. {unexpected[0]}")
    if dreambooth_model_path != "":
        print(f"load dreambooth model from {dreambooth_model_path}")
        ckpt = _load_checkpoint_file(dreambooth_model_path)
        ckpt = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
        pipeline.vae.load_state_dict(convert_ldm_vae_checkpoint(ckpt, pipeline.vae.config))
        pipeline.unet.load_state_dict(convert_ldm_unet_checkpoint(ckpt, pipeline.unet.config), strict=False)
        if getattr(pipeline, "text_encoder", None) is not None:
            te = convert_ldm_clip_checkpoint_concise(ckpt)
            own = pipeline.text_encoder.state_dict()
            # newer transformers no longer register position_ids as a persistent buffer; everything else must match
            te = {k: v for k, v in te.items() if k in own or not k.endswith("position_ids")}
            pipeline.text_encoder.load_state_dict(te, strict=True)
    if lora_model_path != "":
        print(f"load lora model from {lora_model_path}")
        assert lora_model_path.endswith(".safetensors")
        pipeline = convert_lora(pipeline, _load_checkpoint_file(lora_model_path), alpha=lora_alpha)
    if adapter_lora_path != "":
        print(f"load domain lora from {adapter_lora_path}")
        pipeline = load_diffusers_lora(pipeline, _unwrap(_load_checkpoint_file(adapter_lora_path)),
                                       alpha=adapter_lora_scale)
    for cfg in motion_module_lora_configs:
        path, alpha = cfg["path"], cfg["alpha"]
        print(f"load motion LoRA from {path}")
        pipeline = load_diffusers_lora(pipeline, _unwrap(_load_checkpoint_file(path)), alpha)
    return pipeline


def pick_frames(n_total, video_length, fps=None, duration=None):
    """frame indices of util.py:222-230: the first `duration` seconds (or everything), `video_length` evenly spaced picks"""
    total = n_total if duration is None else min(int(fps * duration), n_total)
    return np.linspace(0, total - 1, video_length, dtype=int)


def preprocess_frames(frames_u8, height, width, device=None):
    """util.py:232-238 behind the decoder: uint8 [F, H, W, 3] (numpy or tensor) -> [F, 3, height, width] in [-1, 1];
    resize + normalise are one HIP kernel (`mc_video_resize_u8_f16`)"""
    from .. import ops
    t = torch.as_tensor(frames_u8)
    if device is not None:
        t = t.to(device)
    return ops.video_resize(t, height, width)


def video_preprocess(video_path, height, width, video_length, duration=None, sample_start_idx=0, device=None):
    """reference util.py:217-242: decode -> np.linspace frame pick -> bilinear resize(align_corners=True) -> [-1, 1].
    The decoder is `decord` as in the reference (not part of this image: pass frames to `preprocess_frames`, or video
    latents straight to `obtain_motion_representation(video_latents=...)`)."""
    try:
        import decord
    except ImportError as e:
        raise RuntimeError("video decode needs `decord`; pass decoded frames to preprocess_frames() or video latents "
                           "directly (obtain_motion_representation(video_latents=...))") from e
    vr = decord.VideoReader(video_path)
    idx = pick_frames(len(vr), video_length, vr.get_avg_fps(), duration)
    frames = vr.get_batch(idx)
    frames = frames.asnumpy() if hasattr(frames, "asnumpy") else frames
    dev = device if device is not None else ("cuda" if torch.cuda.is_available() else None)
    return preprocess_frames(frames, height, width, device=dev)
