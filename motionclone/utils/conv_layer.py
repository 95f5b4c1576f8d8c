"""Drop-in alias: `motionclone.utils.conv_layer` resolves to the MI355X-native implementation (same public names as the
reference module motionclone/utils/conv_layer.py)."""
from motionclone_amd.utils.conv_layer import *  # noqa: F401,F403
from motionclone_amd.utils.conv_layer import __dict__ as _d
globals().update({k: v for k, v in _d.items() if not k.startswith("__")})
