"""Which intermediate of tattn_bwd changes first when attention workgroups of another stream share its CUs?
Runs the F = 16, d = 40 backward with the debug dump (mc_tattn_debug_buffer), quiet and under noise, and compares
outputs and intermediates unit by unit.  gpurun: python tools/race_dump.py > gpurun_out/race_dump.log"""
import sys, ctypes, torch
sys.path.insert(0, ".")
from motionclone_amd import lib, ops
dev = torch.device("cuda:0"); L = lib.load()
g = torch.Generator(device=dev).manual_seed(0)
def r(*shape, s=1.0): return (torch.randn(*shape, device=dev, generator=g) * s).half()
HW, H, F, D = 4096, 8, 16, 40
qb = r(F * HW, 960, s=0.5); dob = r(F * HW, 320)
nq = r(F * HW, 960, s=0.5)
ridx = torch.randint(0, 16, (HW, H, F, 1), device=dev, generator=g).to(torch.uint8)
rval = torch.rand((HW, H, F, 1), device=dev, generator=g) * 0.5
units = HW * H
dbg = torch.zeros(units * 64 * 24, device=dev)
NAMES = ["m", "l", "D", "idx", "ref", "PT0", "PT1", "PT2", "PT3", "dPT0", "dPT1", "dPT2", "dPT3", "dsT0", "dsT1", "dsT2",
         "dsT3", "ds0", "ds1", "ds2", "ds3", "pr0", "s0", "dp0"]

def run(mode):
    d = torch.empty_like(qb)
    kw = {} if mode == "dO" else dict(ref_idx=ridx, ref_val=rval, seed_coef=3.0)
    ops.tattn_bwd(qb[:, :320], qb[:, 320:640], qb[:, 640:], None if mode == "seed" else dob, d[:, :320], d[:, 320:640],
                  d[:, 640:], 1, F, HW, H, D, **kw)
    return d
ns = torch.cuda.Stream()
def noise():
    with torch.cuda.stream(ns):
        for _ in range(12): ops.attn_fwd(nq[:, :320], nq[:, 320:640], nq[:, 640:], 4096, 4096, 8, 40, 16)

def unit_mask(a, b, col0):   # [units] bool: any element of the unit differs in the 320-column block at col0
    x = (a[:, col0:col0 + 320] != b[:, col0:col0 + 320]).view(F, HW, H, D)
    return x.any(dim=3).any(dim=0).reshape(-1)

for use_dbg in (True, False):
    L.mc_tattn_debug_buffer(ctypes.c_void_p(dbg.data_ptr() if use_dbg else 0))
    for mode in ("seed", "dO"):
        dbg.zero_()
        ref = run(mode).clone(); torch.cuda.synchronize(); dref = dbg.clone()
        again = run(mode); torch.cuda.synchronize()
        print("[dump=%s mode=%s] quiet repeat identical: out %s, intermediates %s" % (use_dbg, mode, torch.equal(again, ref), torch.equal(dbg, dref)), flush=True)
        hits = 0
        for it in range(10):
            noise(); o = run(mode); torch.cuda.synchronize()
            if torch.equal(o, ref):
                continue
            hits += 1
            mq, mk, mv = unit_mask(o, ref, 0), unit_mask(o, ref, 320), unit_mask(o, ref, 640)
            uq = mq.nonzero().flatten().tolist()
            print("  iter %d: units differing dq %d dk %d dv %d; dq==dk sets: %s" % (it, mq.sum(), mk.sum(), mv.sum(), bool((mq == mk).all())))
            print("    unit ids (first 24):", uq[:24], " min/max", min(uq), max(uq), " blocks", sorted(set(u // 4 for u in uq))[:20])
            if use_dbg:
                A = dbg.view(units, 64, 24); B = dref.view(units, 64, 24)
                fd = (A != B)
                per_field = fd.any(dim=1)          # [units, 24]
                dunits = per_field.any(dim=1)
                print("    units with differing intermediates: %d (subset of dq units: %s; equal: %s)" % (dunits.sum(), bool((dunits & ~mq).sum() == 0), bool((dunits == mq).all())))
                print("    fields differing (count of units):", {NAMES[i]: int(per_field[:, i].sum()) for i in range(24) if per_field[:, i].any()})
                if dunits.any():
                    u = int(dunits.nonzero()[0])
                    lanes = fd[u].any(dim=1).nonzero().flatten().tolist()
                    print("    unit %d: lanes with differences %s" % (u, lanes[:64]))
                    for ln in lanes[:3]:
                        print("      lane %d bad :" % ln, ["%s=%.6g" % (NAMES[i], A[u, ln, i].item()) for i in range(24)])
                        print("      lane %d good:" % ln, ["%s=%.6g" % (NAMES[i], B[u, ln, i].item()) for i in range(24)])
                else:
                    bu = uq[0]; p, h = bu // H, bu % H
                    x = o[:, :320].view(F, HW, H, D)[:, p, h].float(); y = ref[:, :320].view(F, HW, H, D)[:, p, h].float()
                    print("    intermediates identical; dq of unit %d bad/good rows 0-1:" % bu, x[:2, :8].tolist(), y[:2, :8].tolist())
            if hits >= 2:
                break
        print("[dump=%s mode=%s] %d differing runs" % (use_dbg, mode, hits), flush=True)
