"""Import this FIRST in an A/B script: it routes motionclone_amd.lib to the TOOLS build of the kernel library
(tools/_build/libmotionclone_hip_tools.so = the same sources with -DMC_TOOLS), the only build that reads the MC_*
environment switches and exports mc_gemm_debug / mc_gemm_debug_buffer / mc_tattn_debug_buffer.  The product library
(motionclone_amd/csrc/libmotionclone_hip.so) has none of them (tests/test_abi.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from motionclone_amd import build  # noqa: E402

os.environ["MC_HIP_LIB"] = build.build_hip(tools=True)
