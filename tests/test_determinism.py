"""Bit-stability of the temporal-attention backward next to MFMA-heavy kernels of another HIP stream.

Round 2 found `tattn_bwd_kernel` returning a wrong row term D = sum_kv P * dP for a few dozen of 32768 (pixel, head) units
whenever spatial-attention / GEMM waves of ANOTHER stream shared its SIMDs, and only when hipcc's SLP vectoriser had
turned the softmax-backward arithmetic into packed-fp32 (v_pk_*_f32) chains; several videos in flight per GPU
(sampler.sample_interleaved, the bench's default) rest on this kernel being stable.  Three guards:

  * (CPU) the shipped compiler flags leave NO packed-fp32 VALU instruction in any kernel of temporal.hip;
  * (GPU) 100 noisy runs of the shipped library, with and without the dO path: every run bit-identical to the quiet one;
  * (GPU) the same 100 runs on the negative control (the same source built WITH SLP vectorisation,
    motionclone_amd.build.build_slp_control): reported, and expected to differ - it shows that the noise actually reaches
    the failure.  tools/tattn_race.py localises the failing chain (profiles/r03_tattn_race.md).
"""
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_shipped_flags_leave_no_packed_fp32_instructions_in_temporal_kernels():
    from motionclone_amd import build
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "t.s")
        subprocess.run([hipcc] + build.HIP_FLAGS + ["-S", "--cuda-device-only", "-o", out,
                                                    os.path.join(build.CSRC, "temporal.hip")],
                       check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        asm = open(out).read()
    assert "tattn_bwd_kernel" in asm
    packed = re.findall(r"^\s*(v_pk_(?:fma|mul|add)_f32)\b", asm, re.M)
    assert not packed, "%d packed-fp32 instructions in temporal.hip with the shipped flags" % len(packed)


@pytest.mark.gpu
def test_tattn_bwd_is_bit_stable_under_concurrent_mfma_load(gpu_device):
    import json
    import tattn_race as R
    from motionclone_amd import build, lib
    fn = R.bind(lib.HIP_LIB_PATH)
    rows = []
    for seed_only in (False, True):
        bad, worst, units = R.noisy_runs(fn, 100, seed_only)
        rows.append(dict(lib="shipped", seed_only=seed_only, runs=100, differ=bad, worst_units=worst))
        assert bad == 0, "tattn_bwd differs from its quiet result in %d of 100 noisy runs (%d of %d units)" % (bad, worst, units)
    if os.path.exists(build.SLP_CONTROL_LIB):
        ctl = R.bind(build.SLP_CONTROL_LIB)
        for seed_only in (False, True):
            bad, worst, units = R.noisy_runs(ctl, 100, seed_only)
            rows.append(dict(lib="SLP control", seed_only=seed_only, runs=100, differ=bad, worst_units=worst))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "determinism_r03.json"), "w") as f:
        json.dump(rows, f, indent=1)
    print("DETERMINISM", json.dumps(rows))
