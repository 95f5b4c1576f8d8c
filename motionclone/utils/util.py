"""Drop-in alias: `motionclone.utils.util` resolves to the MI355X-native implementation (same public names as the
reference module motionclone/utils/util.py)."""
from motionclone_amd.utils.util import *  # noqa: F401,F403
from motionclone_amd.utils.util import __dict__ as _d
globals().update({k: v for k, v in _d.items() if not k.startswith("__")})
