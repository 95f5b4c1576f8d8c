"""Drop-in alias: `motionclone.models.sparse_controlnet` resolves to the MI355X-native implementation."""
from motionclone_amd.models.sparse_controlnet import *  # noqa: F401,F403
from motionclone_amd.models.sparse_controlnet import SparseControlNetModel  # noqa: F401
