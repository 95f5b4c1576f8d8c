"""Drop-in alias (reference motionclone/utils/convert_lora_safetensor_to_diffusers.py)."""
from motionclone_amd.utils.convert import convert_lora, load_diffusers_lora  # noqa: F401
