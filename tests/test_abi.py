"""The drop-in boundary: every entry point `include/mc_kernels.h` declares is exported by the gfx950 shared library (and by
the host-simulator build of the same sources), and the ctypes table of motionclone_amd/lib.py binds exactly that set.
No compute is called - this runs on GPU-less machines; the HIP library is dlopen'ed in a child process so that this
process never maps a HIP code object before a runtime exists."""
import os
import re
import subprocess
import sys

import pytest

from motionclone_amd import build as B
from motionclone_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header(tools):
    """symbols declared by include/mc_kernels.h: the product part (outside `#ifdef MC_TOOLS`) or the tools-build part"""
    text = open(os.path.join(ROOT, "include", "mc_kernels.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    inside = "".join(re.findall(r"#ifdef MC_TOOLS(.*?)#endif", text, flags=re.S))
    outside = re.sub(r"#ifdef MC_TOOLS.*?#endif", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mc_[a-z0-9_]+)\s*\(", inside if tools else outside)))


def header_symbols():
    return _header(False)


def test_header_and_ctypes_table_agree():
    syms = header_symbols()
    assert len(syms) >= 30 and "mc_gemm_f16" in syms and "mc_version" in syms
    table = set(lib.SIGNATURES) | {"mc_version", "mc_gn_nchunk"}
    assert set(syms) <= table, sorted(set(syms) - table)
    assert set(lib.SIGNATURES) <= set(syms), sorted(set(lib.SIGNATURES) - set(syms))


def test_hip_library_exports_every_declared_symbol():
    path = lib.HIP_LIB_PATH
    if not os.path.exists(path):
        pytest.skip("gfx950 library not built here (python -m motionclone_amd.build)")
    code = ("import ctypes, sys\n"
            "h = ctypes.CDLL(%r)\n"
            "missing = [s for s in %r if not hasattr(h, s)]\n"
            "assert not missing, missing\n"
            "h.mc_version.restype = ctypes.c_int\n"
            "assert h.mc_version() == 1\n"
            "print('ok')\n" % (path, header_symbols()))
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-2000:]


def test_simulator_library_exports_the_same_symbols():
    import ctypes
    h = ctypes.CDLL(B.build_emu())
    missing = [s for s in header_symbols() + _header(True) if not hasattr(h, s)]   # the simulator is a tools build
    assert not missing, missing


def test_product_library_has_no_debug_exports_no_environment_switches_no_mutable_globals():
    """re-entrant, no global mutable state.  The debug hooks and the MC_* A/B switches exist only in builds
    with -DMC_TOOLS (host simulator, tools/_build); the product .so must not export them, must not import getenv, must not
    carry an `MC_` environment name in its string table, and its only writable statics are thread-local."""
    path = lib.HIP_LIB_PATH
    if not os.path.exists(path):
        pytest.skip("gfx950 library not built here (python -m motionclone_amd.build)")
    tools = _header(True)
    assert set(tools) == set(lib.TOOLS_SIGNATURES) and "mc_gemm_debug" in tools
    nm = subprocess.run(["nm", "-D", path], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = {l.split()[-1] for l in nm.splitlines() if " T " in l}
    assert not (exported & set(tools)), exported & set(tools)
    assert not any("debug" in e for e in exported), [e for e in exported if "debug" in e]
    assert set(header_symbols()) <= exported
    assert "getenv" not in nm, "the product library reads the environment"
    raw = open(path, "rb").read()
    names = set(re.findall(rb"MC_[A-Z0-9_]{3,}", raw))
    assert not names, names
    # the library's own writable statics (g_*: launch status, "which kernel ran last") must all be thread-local
    syms = subprocess.run(["readelf", "-sW", path], stdout=subprocess.PIPE, text=True, check=True).stdout
    own = [l.split() for l in syms.splitlines() if re.search(r" _Z(N2mc)?L\d+g_[a-z_]+E?$", l)]
    assert own and all(f[3] == "TLS" for f in own), [f for f in own if f[3] != "TLS"]


def test_product_path_fails_loudly_without_the_library(monkeypatch, tmp_path):
    """no CPU fallback: with the library absent, loading raises KernelLibraryMissing instead of routing elsewhere"""
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "_is_emulated", False)
    monkeypatch.setattr(lib, "HIP_LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(lib.KernelLibraryMissing):
        lib.load()
    import torch
    from motionclone_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):       # CPU tensors are refused outright
        ops.silu(torch.zeros(8, dtype=torch.float16))
