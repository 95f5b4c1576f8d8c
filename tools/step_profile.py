#!/usr/bin/env python
"""Per-(kernel entry point, shape) time model of one guided and one plain DDIM step on the MI355X: every C-ABI call is
bracketed by HIP events on the launch stream (cost: serialises nothing, adds ~2 events per launch), grouped by the
integer / float arguments of the call.  Output: JSON lines sorted by total time + a summary per entry point.

  python tools/step_profile.py [--frames 16 --size 512] [--top 60] > gpurun_out/step_profile.jsonl"""
import argparse
import collections
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from motionclone_amd import lib, ops, spec  # noqa: E402
from motionclone_amd.engine import UNet3DEngine, default_config  # noqa: E402
from motionclone_amd.sampler import MotionCloneSampler  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--top", type=int, default=80)
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()

dev = torch.device("cuda:0")
lib.load()
cfg = default_config()
sd, _ = spec.synthetic_state_dict(cfg, seed=1234, device=dev)
eng = UNet3DEngine(sd, cfg, dev)
smp = MotionCloneSampler(eng)
F, H = args.frames, args.size // 8
g = lambda s: torch.Generator(device=dev).manual_seed(s)   # noqa: E731
lat = torch.randn((1, 4, F, H, H), generator=g(2025), device=dev).half()
text = torch.randn((2, 77, 768), generator=g(7), device=dev).half()
vid = (0.18215 * torch.randn((1, 4, F, H, H), generator=g(11), device=dev)).half()
rep_dev = eng.prepare_representation(smp.extract(vid, lat, text[0:1]))

real_call = lib.call
records = []
on = False


def timed_call(name, *a):
    if not on:
        return real_call(name, *a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    real_call(name, *a)
    e1.record()
    sig = tuple(v for v, t in zip(a, lib.SIGNATURES[name]) if t is not lib.P)
    ptr_mask = tuple(v is not None for v, t in zip(a, lib.SIGNATURES[name]) if t is lib.P)
    records.append((name, sig, ptr_mask, e0, e1))


lib.call = timed_call
ops.lib.call = timed_call


def describe(name, sig, mask):
    if name in ("mc_gemm_f16", "mc_gemm_splitk_f16"):
        M, N, K, lda, lda2, ldc, ldr, c1, ctot, mode, Hs, Ws, Ho, Wo, rpb, alpha, flags = sig[:17]
        d = "mode%d M=%d N=%d K=%d%s%s%s%s" % (mode, M, N, K, " +R" if mask[4] else "", " +A2" if mask[1] else "",
                                              " geglu" if flags & 0x200 else "", " splits=%d" % sig[17] if len(sig) > 17 else "")
        return d, 2.0 * M * N * K
    return " ".join(str(round(v, 4) if isinstance(v, float) else v) for v in sig), 0.0


for tag, idx in (("guided", 0), ("plain", 20)):
    smp.step(lat, idx, text, rep_dev)          # warm
    torch.cuda.synchronize()
    records.clear()
    on = True
    for _ in range(args.reps):
        smp.step(lat, idx, text, rep_dev)
    on = False
    torch.cuda.synchronize()
    groups = collections.OrderedDict()
    for name, sig, mask, e0, e1 in records:
        k = (name, sig, mask)
        gg = groups.setdefault(k, [0, 0.0])
        gg[0] += 1
        gg[1] += e0.elapsed_time(e1)
    total = sum(v[1] for v in groups.values()) / args.reps
    byname = collections.OrderedDict()
    rows = []
    for (name, sig, mask), (n, ms) in groups.items():
        d, fl = describe(name, sig, mask)
        n, ms = n / args.reps, ms / args.reps
        rows.append(dict(step=tag, kernel=name[3:], shape=d, launches=n, ms=ms, share=ms / total, avg_us=1e3 * ms / n,
                         tflops=(fl * n / ms / 1e9) if fl else None))
        b = byname.setdefault(name[3:], [0, 0.0])
        b[0] += n
        b[1] += ms
    rows.sort(key=lambda r: -r["ms"])
    print(json.dumps(dict(step=tag, total_event_ms=total, launches=sum(r["launches"] for r in rows),
                          by_kernel={k: dict(launches=v[0], ms=round(v[1], 3), share=round(v[1] / total, 4))
                                     for k, v in sorted(byname.items(), key=lambda kv: -kv[1][1])})))
    for r in rows[:args.top]:
        print(json.dumps(r))
