"""Issue-rate probes on cuda:0 (tools/issue_rate.hip): cycles per wave64 instruction of the VALU kinds the attention softmax uses,
of the 16x16x32 MFMA, and how much of a VALU stream overlaps an MFMA stream of ANOTHER wave on the same SIMD vs of the same wave."""
import ctypes
import json
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "_build", "libissue_rate.so")


def build():
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so,
                           os.path.join(here, "issue_rate.hip")])


if "--build" in sys.argv:
    build()
    sys.exit(0)
lib = ctypes.CDLL(so)
NAMES = {0: "idle", 1: "fma_f32", 2: "exp_f32", 3: "cvt_pkrtz", 4: "max3_f32", 5: "mul_f32", 6: "exp_f16", 7: "pk_fma_f16",
         8: "pk_mul_f32", 10: "mfma16x16x32", 21: "16 mfma + 64 mul, one wave", 22: "16 mfma + 64 exp, one wave",
         23: "16 mfma + 4 mul + 16 mfma, one wave"}
ITERS = 2000


def run(a, b, nw):
    cy = (ctypes.c_uint64 * 8)()
    ms = ctypes.c_float()
    rc = lib.run_probe(a, b, nw, ITERS, cy, ctypes.byref(ms))
    assert rc == 0, rc
    return [int(c) for c in cy[:nw]], ms.value


rows = []
for k in (1, 5, 2, 3, 4, 6, 7, 8, 10, 21, 22, 23):
    cy, ms = run(k, 0, 4)
    per = {1: 64, 5: 64, 2: 64, 3: 64, 4: 64, 6: 64, 7: 64, 8: 64, 10: 64, 21: 80, 22: 80, 23: 36}[k]
    rows.append(dict(test=NAMES[k], waves_per_simd=1, cycles_per_trip=cy[0] / ITERS, cycles_per_instr=cy[0] / ITERS / per, ms=ms))
# two waves per SIMD: same kind twice (does the pipe rate hold), then MFMA beside VALU kinds
for a, b in ((1, 1), (2, 2), (10, 10), (10, 5), (10, 2), (10, 3), (10, 4), (5, 2)):
    cy, ms = run(a, b, 8)
    rows.append(dict(test="%s || %s" % (NAMES[a], NAMES[b]), waves_per_simd=2, cycles_per_trip_a=cy[0] / ITERS,
                     cycles_per_trip_b=cy[4] / ITERS, ms=ms))
for r in rows:
    print(json.dumps(r))
