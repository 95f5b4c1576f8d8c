"""Drop-in alias: `motionclone.models.unet` resolves to the MI355X-native implementation (same public names as the
reference module motionclone/models/unet.py)."""
from motionclone_amd.models.unet import *  # noqa: F401,F403
from motionclone_amd.models.unet import __dict__ as _d
globals().update({k: v for k, v in _d.items() if not k.startswith("__")})
