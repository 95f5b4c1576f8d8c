"""Drop-in alias: the VAE the reference imports from diffusers (`from diffusers import AutoencoderKL`,
t2v_video_sample.py:4), MI355X-native decode path."""
from motionclone_amd.models.vae import *  # noqa: F401,F403
from motionclone_amd.models.vae import AutoencoderKL, DecoderOutput  # noqa: F401
