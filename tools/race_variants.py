"""tattn_bwd under attention noise from another stream, for the exchange variants of temporal.hip
(MC_TATTN_VARIANT: unset = ds_bpermute, 2 = + lgkmcnt(0) drain, 3 = v_permlane*_swap)."""
import os, sys, torch
sys.path.insert(0, ".")
from motionclone_amd import lib, ops
dev = torch.device("cuda:0"); lib.load()
if os.environ.get("MC_ALT_LIB"):   # e.g. a build of temporal.hip with other compiler flags
    lib._lib = lib._bind(os.environ["MC_ALT_LIB"]); lib._FN.clear()
    print("using", os.environ["MC_ALT_LIB"])
g = torch.Generator(device=dev).manual_seed(0)
def r(*shape, s=1.0): return (torch.randn(*shape, device=dev, generator=g) * s).half()
HW, H, F, D = 4096, 8, 16, 40
qb = r(F * HW, 960, s=0.5); dob = r(F * HW, 320); nq = r(F * HW, 960, s=0.5)
ridx = torch.randint(0, 16, (HW, H, F, 1), device=dev, generator=g).to(torch.uint8)
rval = torch.rand((HW, H, F, 1), device=dev, generator=g) * 0.5
def run():
    d = torch.empty_like(qb)
    ops.tattn_bwd(qb[:, :320], qb[:, 320:640], qb[:, 640:], dob, d[:, :320], d[:, 320:640], d[:, 640:], 1, F, HW, H, D,
                  ref_idx=ridx, ref_val=rval, seed_coef=3.0)
    return d
ns = torch.cuda.Stream()
def noise():
    with torch.cuda.stream(ns):
        for _ in range(12): ops.attn_fwd(nq[:, :320], nq[:, 320:640], nq[:, 640:], 4096, 4096, 8, 40, 16)
base = None
for var in ("", "4", ""):
    if var: os.environ["MC_TATTN_VARIANT"] = var
    else: os.environ.pop("MC_TATTN_VARIANT", None)
    ref = run().clone(); torch.cuda.synchronize()
    if base is None: base = ref
    bad = 0; n = 12
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    for _ in range(n):
        noise(); o = run(); torch.cuda.synchronize(); bad += int(not torch.equal(o, ref))
    print("variant %-2s: quiet result == variant-0 result: %s; %.1f us per call; under noise %d/%d runs differ"
          % (var or "0", torch.equal(ref, base), 1e3 * e0.elapsed_time(e1) / 20, bad, n), flush=True)
