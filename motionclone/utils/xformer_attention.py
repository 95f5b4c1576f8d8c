"""Drop-in alias: `motionclone.utils.xformer_attention` resolves to the MI355X-native implementation (same public names as the
reference module motionclone/utils/xformer_attention.py)."""
from motionclone_amd.utils.xformer_attention import *  # noqa: F401,F403
from motionclone_amd.utils.xformer_attention import __dict__ as _d
globals().update({k: v for k, v in _d.items() if not k.startswith("__")})
