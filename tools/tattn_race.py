"""tattn_bwd_kernel next to MFMA-heavy kernels of another stream: is it bit-stable, and which code makes it unstable?

One tool for the whole investigation (round 2's race_hunt*.py / race_variants.py / race_dump.py collapsed):

  python tools/tattn_race.py build            # here, no GPU: variant libraries of csrc/temporal.hip -> tools/_build/
  python tools/tattn_race.py run [--runs N]   # on the MI355X: every variant, N noisy runs each, one JSON line per variant
  python tools/tattn_race.py isa              # here: packed-fp32 instruction census of the backward kernel per variant

Variants (all from the SAME source, csrc/temporal.hip):
  product        the shipped flags (-fno-slp-vectorize): must be bit-stable
  slp            hipcc's SLP vectoriser on = round 2's failing build (negative control: proves the noise is sensitive)
  slp_pv / slp_d / slp_sum / slp_all
                 SLP on, but the value named is made opaque to the optimiser (MC_TATTN_PROBE bits 1 / 2 / 4 / 7) so
                 that no packed-fp32 instruction can be formed ACROSS it: localises the failing chain of
                 D = sum_kv P * dP  (P = e / l;  dP += coef * (P - ref) at the seeded kv;  D += P * dP).

A run = the F = 16, d = 40, 4096-pixel backward (config-2 level-0 shape of up_blocks.1's neighbour) quiet, then N times
with 12 level-0 spatial-attention forwards in flight on a second stream; a run "differs" if any of dq / dk / dv differs
bitwise from the quiet result; the number of differing (pixel, head) units is reported for the worst run."""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _toolslib  # noqa: E401,E702,F401  (tools build of the library: MC_* switches / debug hooks)
import argparse
import ctypes
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
OUT = os.path.join(HERE, "_build")
SRC = os.path.join(REPO, "motionclone_amd", "csrc", "temporal.hip")
BASE = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffast-math", "-fno-finite-math-only"]
VARIANTS = {
    "product": ["-fno-slp-vectorize"],
    "slp": [],
    "slp_pv": ["-DMC_TATTN_PROBE=1"],
    "slp_d": ["-DMC_TATTN_PROBE=2"],
    "slp_sum": ["-DMC_TATTN_PROBE=4"],
    "slp_all": ["-DMC_TATTN_PROBE=7"],
}
KERNEL = "_ZN2mc16tattn_bwd_kernelILi1ELi3ELi0E"


def lib_path(name):
    return os.path.join(OUT, "libtattn_%s.so" % name)


def build(emit_asm=False):
    os.makedirs(OUT, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    for name, extra in VARIANTS.items():
        cmd = [hipcc] + BASE + extra
        if emit_asm:
            cmd += ["-S", "--cuda-device-only", "-o", os.path.join(OUT, "tattn_%s.s" % name), SRC]
        else:
            cmd += ["-shared", "-o", lib_path(name), SRC]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode:
            raise SystemExit(r.stdout)
        print("built", name)


def isa():
    build(emit_asm=True)
    for name in VARIANTS:
        txt = open(os.path.join(OUT, "tattn_%s.s" % name)).read()
        m = re.search(r"^%s[^\n]*\n(.*?)\.amdhsa_kernel" % KERNEL, txt, re.S | re.M)
        body = m.group(1)
        pk = re.findall(r"\bv_pk_(fma|mul|add)_f32\b", body)
        lines = body.split("\n")
        # what feeds the first exchange of the D reduction (third v_permlane16_swap of the kernel: max, sum, D)
        idx = [i for i, ln in enumerate(lines) if "v_permlane16_swap" in ln]
        feed = [ln.strip() for ln in lines[max(0, idx[2] - 8):idx[2] + 1]] if len(idx) >= 3 else []
        print(json.dumps({"variant": name, "v_pk_f32_in_bwd_kernel": len(pk),
                          "by_op": {o: pk.count(o) for o in ("fma", "mul", "add")},
                          "instructions_before_D_exchange": feed}))


def bind(path):
    lib = ctypes.CDLL(path)
    from motionclone_amd.lib import SIGNATURES
    fn = lib.mc_tattn_bwd_f16
    fn.argtypes = SIGNATURES["mc_tattn_bwd_f16"]
    fn.restype = ctypes.c_int
    return fn


def noisy_runs(fn, runs, seed_only=False, report=None):
    """-> (runs that differ, worst number of differing (pixel, head) units, units)"""
    import torch
    from motionclone_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)

    def r(*shape, s=1.0):
        return (torch.randn(*shape, device=dev, generator=g) * s).half()
    HW, H, F, D = 4096, 8, 16, 40
    qb, dob, nq = r(F * HW, 960, s=0.5), r(F * HW, 320), r(F * HW, 960, s=0.5)
    ridx = torch.randint(0, 16, (HW, H, F, 1), device=dev, generator=g).to(torch.uint8)
    rval = torch.rand((HW, H, F, 1), device=dev, generator=g) * 0.5

    def run():
        d = torch.zeros_like(qb)
        st = torch.cuda.current_stream().cuda_stream
        rc = fn(qb.data_ptr(), qb[:, 320:].data_ptr(), qb[:, 640:].data_ptr(), 960,
                None if seed_only else dob.data_ptr(), 320, d.data_ptr(), d[:, 320:].data_ptr(), d[:, 640:].data_ptr(), 960,
                ridx.data_ptr(), rval.data_ptr(), 3.0, 1, F, HW, H, D, D ** -0.5, st)
        assert rc == 0, rc
        return d
    ns = torch.cuda.Stream()

    def noise():
        with torch.cuda.stream(ns):
            for _ in range(12):
                ops.attn_fwd(nq[:, :320], nq[:, 320:640], nq[:, 640:], 4096, 4096, 8, 40, 16)
    ref = run().clone()
    torch.cuda.synchronize()
    assert torch.equal(run(), ref), "not even reproducible on a quiet GPU"
    bad = worst = 0
    for _ in range(runs):
        noise()
        o = run()
        torch.cuda.synchronize()
        if not torch.equal(o, ref):
            bad += 1
            diff = (o != ref).view(F, HW, 3, H, D).any(-1).any(2).any(0)   # [HW, H]
            worst = max(worst, int(diff.sum()))
    return bad, worst, HW * H


def run(runs):
    import torch
    from motionclone_amd import lib
    lib.load()
    res = []
    for name in VARIANTS:
        fn = bind(lib_path(name))
        for seed_only in (False, True):
            bad, worst, units = noisy_runs(fn, runs, seed_only)
            row = {"variant": name, "mode": "seed only (dO = null)" if seed_only else "dO + seed", "noisy_runs": runs,
                   "runs_that_differ": bad, "worst_units_differing": worst, "units": units,
                   "gpu": torch.cuda.get_device_name(0)}
            print(json.dumps(row), flush=True)
            res.append(row)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["build", "run", "isa"])
    ap.add_argument("--runs", type=int, default=100)
    a = ap.parse_args()
    {"build": build, "isa": isa, "run": lambda: run(a.runs)}[a.cmd]()
