"""Build helpers for libmotionclone_hip.so (gfx950) and, for tests only, the host simulator build.

`build_hip()` is what `__graft_entry__.build()` calls: one hipcc invocation per .hip source
(cross-compiles without a GPU), linked in-tree so the .so travels with the repo snapshot.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
REPO = os.path.dirname(HERE)
SOURCES = ["gemm_api.hip", "gemm2.hip", "gemm3.hip", "gemm4.hip", "gemm5.hip", "gemm6.hip", "norm.hip", "elementwise.hip", "temporal.hip", "attention.hip"]
HIP_LIB = os.path.join(CSRC, "libmotionclone_hip.so")
EMU_DIR = os.path.join(REPO, "tests", "hipemu")
EMU_LIB = os.path.join(EMU_DIR, "_build", "libmc_emu.so")


# -fno-slp-vectorize: no packed fp32 VALU instructions (v_pk_fma_f32 ...).  Measured: the temporal-attention backward is
# not bit-reproducible with them when MFMA waves of another stream share its SIMDs (csrc/temporal.hip header,
# tools/tattn_race.py, tests/test_determinism.py), and the library is 1.4 % faster end to end without them (a v_pk_fma_f32
# costs two v_fma_f32 issue slots beside MFMAs anyway).
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffast-math", "-fno-finite-math-only",
             "-fno-slp-vectorize"]
# gemm6.hip: hipcc's atomic optimizer turns the tile counter's one-lane atomicAdd into a wave reduction whose result is read (and
# waited for with s_waitcnt vmcnt(0): a drain of the operand ring) right behind the atomic; left alone, the compiler waits for
# the value where the source reads it - behind the epilogue - with an exact count
FILE_FLAGS = {"gemm6.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]}
# TEST ONLY: csrc/temporal.hip with the SLP vectoriser ON = the build whose backward went wrong in round 2; the negative
# control of tests/test_determinism.py (never loaded by the package)
SLP_CONTROL_LIB = os.path.join(REPO, "tools", "_build", "libtattn_slp.so")


def _stamp(paths, extra=""):
    h = hashlib.sha1(extra.encode())
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def _deps():
    import glob
    return [os.path.join(CSRC, s) for s in SOURCES] + sorted(glob.glob(os.path.join(CSRC, "*.hpp")))


TOOLS_LIB = os.path.join(REPO, "tools", "_build", "libmotionclone_hip_tools.so")


OLD_EPI_LIB = os.path.join(REPO, "tools", "_build", "libmotionclone_hip_oldepi.so")


def build_old_epilogue_control(force=False):
    """A/B ONLY (bench.py through MC_HIP_LIB): the product sources with gemm5's one-pass kernels on round 5's epilogue
    (-DMC_G5_OLD_EPILOGUE) - what the round-6 epilogue of the convolutions is measured against."""
    return build_hip(force=force, out_lib=OLD_EPI_LIB, extra=["-DMC_G5_OLD_EPILOGUE"], objdir=os.path.join(REPO, "tools", "_build", "obj_oldepi"))


def build_hip(force=False, verbose=False, tools=False, out_lib=None, extra=(), objdir=None):
    """Compile every HIP source for gfx950 and link csrc/libmotionclone_hip.so.

    tools=True: the same sources with -DMC_TOOLS -> tools/_build/libmotionclone_hip_tools.so, the library the A/B scripts
    under tools/ load through MC_HIP_LIB: it reads the MC_* environment switches and exports mc_gemm_debug* /
    mc_tattn_debug_buffer.  The product library (tools=False) has no environment read and no mutable global."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = HIP_FLAGS + (["-DMC_TOOLS"] if tools else []) + list(extra)
    out_lib = out_lib or (TOOLS_LIB if tools else HIP_LIB)
    stamp = _stamp(_deps(), " ".join(flags) + repr(sorted(FILE_FLAGS.items())))
    stamp_file = out_lib + ".stamp"
    if not force and os.path.exists(out_lib) and os.path.exists(stamp_file):
        if open(stamp_file).read() == stamp:
            return out_lib
    objdir = objdir or (os.path.join(os.path.dirname(TOOLS_LIB), "obj") if tools else os.path.join(CSRC, "build"))
    os.makedirs(objdir, exist_ok=True)

    def one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        out = _run([hipcc] + flags + FILE_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj])
        if verbose and out.strip():
            print(out)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(one, SOURCES))
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out_lib] + objs)
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return out_lib


def build_slp_control(force=False):
    """TEST ONLY: the negative control of the determinism stress test (temporal.hip with SLP vectorisation)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    src = os.path.join(CSRC, "temporal.hip")
    flags = [f for f in HIP_FLAGS if f != "-fno-slp-vectorize"] + ["-DMC_TOOLS"]
    stamp = _stamp([src, os.path.join(CSRC, "mc_common.hpp")], " ".join(flags))
    stamp_file = SLP_CONTROL_LIB + ".stamp"
    if not force and os.path.exists(SLP_CONTROL_LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return SLP_CONTROL_LIB
    os.makedirs(os.path.dirname(SLP_CONTROL_LIB), exist_ok=True)
    _run([hipcc] + flags + ["-shared", "-o", SLP_CONTROL_LIB, src])
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return SLP_CONTROL_LIB


# TEST ONLY: csrc/attention.hip with the round-5 fix of scale_in_place left out (the inline v_mul reads v_exp_f32's result one
# cycle too early): the negative control of tests/test_kernels.py::test_attention_rebase_negative_control (never loaded by the package)
TRANS_HAZARD_CONTROL_LIB = os.path.join(REPO, "tools", "_build", "libattn_trans_hazard.so")


def build_trans_hazard_control(force=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    src = os.path.join(CSRC, "attention.hip")
    flags = HIP_FLAGS + ["-DMC_TOOLS", "-DMC_CONTROL_TRANS_HAZARD"]
    stamp = _stamp([src] + [d for d in _deps() if d.endswith(".hpp")], " ".join(flags))
    stamp_file = TRANS_HAZARD_CONTROL_LIB + ".stamp"
    if (not force and os.path.exists(TRANS_HAZARD_CONTROL_LIB) and os.path.exists(stamp_file)
            and open(stamp_file).read() == stamp):
        return TRANS_HAZARD_CONTROL_LIB
    os.makedirs(os.path.dirname(TRANS_HAZARD_CONTROL_LIB), exist_ok=True)
    _run([hipcc] + flags + ["-shared", "-o", TRANS_HAZARD_CONTROL_LIB, src])
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return TRANS_HAZARD_CONTROL_LIB


def build_emu(force=False):
    """TEST ONLY: compile the same kernel sources for the host against tests/hipemu (no GPU needed)."""
    cxx = os.environ.get("MC_HOST_CXX", "/opt/rocm/lib/llvm/bin/clang++")
    deps = _deps() + [os.path.join(EMU_DIR, "hip_emu.h"), os.path.join(EMU_DIR, "hip_emu.cpp")]
    flags = ["-std=c++20", "-O2", "-mf16c", "-DMC_EMU", "-fPIC", "-I", EMU_DIR, "-I", CSRC,
             "-Wno-unused-value", "-Wno-pass-failed"]
    stamp = _stamp(deps, " ".join(flags))
    stamp_file = EMU_LIB + ".stamp"
    if not force and os.path.exists(EMU_LIB) and os.path.exists(stamp_file):
        if open(stamp_file).read() == stamp:
            return EMU_LIB
    os.makedirs(os.path.dirname(EMU_LIB), exist_ok=True)
    objdir = os.path.dirname(EMU_LIB)

    def one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        lang = ["-x", "c++"] if src.endswith(".hip") else []
        _run([cxx] + flags + lang + ["-c", src, "-o", obj])
        return obj

    srcs = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(EMU_DIR, "hip_emu.cpp")]
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(one, srcs))
    _run([cxx, "-shared", "-fPIC", "-o", EMU_LIB] + objs + ["-lpthread"])
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return EMU_LIB


if __name__ == "__main__":
    if "--emu" in sys.argv:
        print(build_emu(force="--force" in sys.argv))
    elif "--tools" in sys.argv:
        print(build_hip(force="--force" in sys.argv, verbose=True, tools=True))
    else:
        print(build_hip(force="--force" in sys.argv, verbose=True))
