#!/usr/bin/env python
"""Launch inventory of one guided / plain step at full size WITHOUT running kernels: lib.call is replaced by a recorder,
tensors are uninitialised CPU buffers.  -> JSON: per kernel entry point and shape, count / flops / algorithmic bytes.
  python tools/launch_inventory.py [--frames 16 --size 512]"""
import argparse
import collections
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from motionclone_amd import build, lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--size", type=int, default=512)
args = ap.parse_args()
lib.use_library_for_tests(build.build_emu())
calls = []
lib.call = lambda name, *a: calls.append((name, a))
ops.lib.call = lib.call

from motionclone_amd import spec  # noqa: E402
from motionclone_amd.engine import UNet3DEngine, default_config  # noqa: E402
from motionclone_amd.sampler import MotionCloneSampler  # noqa: E402

cfg = default_config()
shapes = spec.param_shapes(cfg)
sd = {k: torch.empty(s, dtype=torch.float16) for k, s in shapes.items()}
eng = UNet3DEngine(sd, cfg, "cpu")
smp = MotionCloneSampler(eng)
F, H = args.frames, args.size // 8
lat = torch.empty(1, 4, F, H, H, dtype=torch.float16)
text = torch.empty(2, 77, 768, dtype=torch.float16)
rep = {n: (torch.empty(H * H // 16, 8, F, 1, dtype=torch.uint8), torch.empty(H * H // 16, 8, F, 1)) for n in eng.hooked_names()}


def summarize(tag):
    groups = collections.OrderedDict()
    for name, a in calls:
        if name in ("mc_gemm_f16", "mc_gemm_splitk_f16"):
            M, N, K, mode = a[6], a[7], a[8], a[15]
            key = "%s mode%d M=%d N=%d K=%d%s%s" % (name[3:-4], mode, M, N, K, " +R" if a[4] else "", " geglu" if a[22] & 0x200 else "")
            fl = 2.0 * M * N * K
        else:
            key = name[3:]
            fl = 0
        g = groups.setdefault(key, [0, 0.0])
        g[0] += 1
        g[1] += fl
    print("== %s: %d launches, %.2f TFLOP in GEMMs" % (tag, len(calls), sum(g[1] for g in groups.values()) / 1e12))
    for k, (n, fl) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
        print("%5d  %8.3f TF  %s" % (n, fl / 1e12, k))
    calls.clear()


smp.step(lat, 0, text, rep)
summarize("guided step")
smp.step(lat, 20, text, rep)
summarize("plain step")
