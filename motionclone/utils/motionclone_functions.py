"""Drop-in alias: `motionclone.utils.motionclone_functions` resolves to the MI355X-native implementation (same public names as the
reference module motionclone/utils/motionclone_functions.py)."""
from motionclone_amd.utils.motionclone_functions import *  # noqa: F401,F403
from motionclone_amd.utils.motionclone_functions import __dict__ as _d
globals().update({k: v for k, v in _d.items() if not k.startswith("__")})
