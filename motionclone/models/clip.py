"""Drop-in alias: the text encoder the reference takes from transformers (`CLIPTextModel`, t2v_video_sample.py:5,24)."""
from motionclone_amd.models.clip import CLIPTextModel, clip_param_shapes  # noqa: F401
