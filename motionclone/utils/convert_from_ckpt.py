"""Drop-in alias for the converters `load_weights` imports (reference motionclone/utils/convert_from_ckpt.py)."""
from motionclone_amd.utils.convert import (convert_ldm_clip_checkpoint_concise, convert_ldm_unet_checkpoint,  # noqa: F401
                                           convert_ldm_vae_checkpoint)
