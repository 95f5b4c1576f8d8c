"""Static guard for the hazard class that cost the ring attention its re-base path (round 5): on gfx940+ a VALU
instruction that reads the result of a transcendental (v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos) needs one wait
state in between; hipcc inserts it for its own instructions but does not look inside inline asm.  Every kernel source is
compiled to gfx950 ISA and scanned: no inline-asm instruction may read a TRANS result within the next two issue slots unless an
s_nop sits in front of it.  The scanner is checked against the negative control (attention.hip built with
-DMC_CONTROL_TRANS_HAZARD = the pre-fix code), which it must flag.  CPU-only: hipcc cross-compiles without a GPU."""
import os
import re
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

from motionclone_amd import build

TRANS = re.compile(r"^\s*(v_(?:exp|log|rcp|rsq|sqrt|sin|cos)_f(?:16|32)|v_rcp_iflag_f32)\S*\s+(v\d+|v\[\d+:\d+\])")


def scan(asm):
    """-> [(line number, trans instruction, inline-asm consumer)]"""
    lines = asm.split("\n")
    hits = []
    for i, ln in enumerate(lines):
        m = TRANS.match(ln)
        if not m:
            continue
        dst = m.group(2)
        j, slots, in_asm = i + 1, 0, False
        while j < len(lines) and slots < 2:
            t = lines[j].strip()
            j += 1
            if "ASMSTART" in t:
                in_asm = True
                continue
            if "ASMEND" in t:
                in_asm = False
                continue
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                continue
            if t.startswith("s_nop"):
                break                      # a wait state: whatever follows is safe
            if in_asm and re.search(r"\b" + re.escape(dst) + r"\b", t):
                hits.append((i + 1, ln.strip(), t))
            slots += 1
    return hits


MFMA = re.compile(r"^\s*v_mfma_\S+\s+v\[(\d+):(\d+)\]")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def scan_mfma_to_asm(asm, window=12):
    """XDL (MFMA) write of a VGPR followed by a VALU read needs up to 11 - 19 wait states on gfx940+ (software-inserted: hipcc does it
    for its own instructions): an inline-asm VALU instruction must not read a register an MFMA wrote within the last `window`
    instructions.  -> [(line number, mfma, consumer)]"""
    lines = asm.split("\n")
    real = []      # (line number, text, in_asm)
    in_asm = False
    for i, ln in enumerate(lines):
        t = ln.strip()
        if "ASMSTART" in t:
            in_asm = True
            continue
        if "ASMEND" in t:
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        real.append((i + 1, t, in_asm))
    hits = []
    for k, (no, t, ia) in enumerate(real):
        if not ia or not t.startswith("v_"):
            continue
        ops = t.split(None, 1)[1] if " " in t else ""
        srcs = set()
        for m in VREG.finditer(ops):
            if m.group(1) is not None:
                srcs.add(int(m.group(1)))
            else:
                srcs.update(range(int(m.group(2)), int(m.group(3)) + 1))
        waited = 0
        for back in range(k - 1, max(-1, k - 1 - window), -1):
            bt = real[back][1]
            if bt.startswith("s_nop"):
                mm = re.match(r"s_nop\s+(\d+)", bt)
                waited += int(mm.group(1)) + 1 if mm else 1
            if waited + (k - 1 - back) >= window:
                break
            m = MFMA.match(bt)
            if m and srcs & set(range(int(m.group(1)), int(m.group(2)) + 1)):
                hits.append((no, bt, t))
                break
    return hits


def isa(src, extra=()):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run([hipcc] + build.HIP_FLAGS + list(extra) + ["-S", "--cuda-device-only", "-o", out, os.path.join(build.CSRC, src)],
                       check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        return open(out).read()


@pytest.fixture(scope="module")
def have_hipcc():
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")


def test_no_inline_asm_reads_a_transcendental_result_without_a_wait_state(have_hipcc):
    with ThreadPoolExecutor(max_workers=min(8, len(build.SOURCES))) as ex:
        texts = list(ex.map(isa, build.SOURCES))
    total_asm = 0
    for src, text in zip(build.SOURCES, texts):
        total_asm += text.count("ASMSTART")
        hits = scan(text)
        assert not hits, "%s: inline asm reads a TRANS result too early: %s" % (src, hits[:4])
        hits = scan_mfma_to_asm(text)
        assert not hits, "%s: inline asm reads an MFMA result inside the XDL-write wait window: %s" % (src, hits[:4])
    assert total_asm > 500          # the scan really saw the inline-asm sites (attention, gemm4, gemm5, temporal ...)


def test_the_scanner_flags_the_pre_fix_code(have_hipcc):
    hits = scan(isa("attention.hip", ["-DMC_TOOLS", "-DMC_CONTROL_TRANS_HAZARD"]))
    assert len(hits) >= 4 and all("v_mul_f32" in h[2] and "v_exp_f32" in h[1] for h in hits), hits[:4]
