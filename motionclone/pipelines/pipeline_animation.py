"""Drop-in alias: `motionclone.pipelines.pipeline_animation` resolves to the MI355X-native implementation (same public names as the
reference module motionclone/pipelines/pipeline_animation.py)."""
from motionclone_amd.pipelines.pipeline_animation import *  # noqa: F401,F403
from motionclone_amd.pipelines.pipeline_animation import __dict__ as _d
globals().update({k: v for k, v in _d.items() if not k.startswith("__")})
