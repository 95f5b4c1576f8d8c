"""Host-side helpers of the path (reference motionclone/utils/util.py).  Only the pieces the hot path touches are
provided; weight-file conversion (convert_from_ckpt / LoRA merge) and video decode are listed as "next" in
SURVEY.md 8f and raise a clear error here."""
import random

import numpy as np
import torch


def classify_blocks(block_list, name):
    """substring match of a module name against the configured guidance blocks (reference util.py:434-440)"""
    return any(block in name for block in block_list)


def set_all_seed(seed):
    """reference util.py:442-447"""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def load_weights(pipeline, motion_module_path="", dreambooth_model_path="", **unused):
    """reference util.py:115-215.  Loads an AnimateDiff motion-module checkpoint (keys containing
    'motion_modules.') into the UNet; DreamBooth / LoRA conversion needs the LDM->diffusers key maps, which are
    load-time host code outside the hot path."""
    if motion_module_path:
        sd = torch.load(motion_module_path, map_location="cpu")
        sd = sd["state_dict"] if "state_dict" in sd else sd
        sd = {k: v for k, v in sd.items() if "motion_modules." in k and "pos_encoder.pe" not in k}
        missing, unexpected = pipeline.unet.load_state_dict(sd, strict=False)
        assert len(unexpected) == 0, unexpected
        print(f"load motion module from {motion_module_path}")
    if dreambooth_model_path:
        raise NotImplementedError("DreamBooth/LoRA checkpoint conversion is not part of the hot path (SURVEY.md 8f)")
    return pipeline


def video_preprocess(video_path, height, width, video_length, duration=None, sample_start_idx=0):
    """reference util.py:217-242: decode -> np.linspace frame pick -> bilinear resize(align_corners=True) -> [-1, 1].
    Needs decord, which is not part of this image; synthetic latents bypass it."""
    try:
        import decord  # noqa: F401
    except ImportError as e:
        raise RuntimeError("video decode needs `decord`; pass video latents directly "
                           "(obtain_motion_representation(video_latents=...))") from e
    vr = decord.VideoReader(video_path)
    fps = vr.get_avg_fps()
    total = len(vr) if duration is None else min(int(duration * fps), len(vr))
    idx = np.linspace(sample_start_idx, total - 1, video_length, dtype=int)
    frames = torch.from_numpy(vr.get_batch(idx).asnumpy()).permute(0, 3, 1, 2).float()
    frames = torch.nn.functional.interpolate(frames, size=(height, width), mode="bilinear", align_corners=True)
    return frames / 127.5 - 1.0
