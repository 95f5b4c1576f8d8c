"""Experiment (round 4): do three videos in flight run faster when they are in DIFFERENT phases of the schedule (guided steps
with their backward next to plain steps) than when all three walk the schedule in step?  Same 90 graph replays either way:
  in phase:  for i in 0..29: lane k replays step i
  rotated:   for g in 0..29: lane k replays step (g + 10 k) mod 30
(the latents fed to a step are whatever the lane's previous replay left - timing only).  One JSON line.
  python tools/lane_phase_ab.py > gpurun_out/r04_lane_phase_ab.json"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motionclone_amd import lib, ops, spec  # noqa: E402
from motionclone_amd.engine import UNet3DEngine, default_config  # noqa: E402
from motionclone_amd.sampler import MotionCloneSampler  # noqa: E402

lib.load()
dev = torch.device("cuda", 0)
cfg = default_config()
sd, _ = spec.synthetic_state_dict(cfg, seed=1234, device=dev)
eng = UNet3DEngine(sd, cfg, dev)
NL, N = 3, 30
ops.set_gemm_share(NL)
g = torch.Generator(device=dev).manual_seed(3)
streams = [torch.cuda.Stream() for _ in range(NL)]
smps = [MotionCloneSampler(eng, num_inference_steps=N, guidance_steps=18, guidance_scale=0.4).enable_graphs() for _ in range(NL)]
lats = [torch.randn((1, 4, 16, 64, 64), generator=g, device=dev, dtype=torch.float16) for _ in range(NL)]
text = torch.randn((2, 77, 768), generator=g, device=dev).half()
vid = (0.18215 * torch.randn((1, 4, 16, 64, 64), generator=g, device=dev)).half()
rep = eng.prepare_representation(smps[0].extract(vid, lats[0], text[0:1]))
for k in range(NL):                      # capture all 30 graphs of every lane (one lane at a time)
    with torch.cuda.stream(streams[k]):
        x = lats[k]
        for i in range(N):
            x = smps[k].step(x, i, text, rep)
    torch.cuda.synchronize()


def run(order, reps=3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for gstep in range(N):
            for k in range(NL):
                with torch.cuda.stream(streams[k]):
                    smps[k].step(lats[k], order(gstep, k), text, rep)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


res = {}
for name, order in [("in_phase", lambda s, k: s), ("rotated", lambda s, k: (s + 10 * k) % N), ("in_phase_2", lambda s, k: s),
                    ("rotated_2", lambda s, k: (s + 10 * k) % N)]:
    run(order, 1)
    res[name] = run(order)
res["videos_per_min_in_phase"] = 60.0 * NL / min(res["in_phase"], res["in_phase_2"])
res["videos_per_min_rotated"] = 60.0 * NL / min(res["rotated"], res["rotated_2"])
res["note"] = "30 graph replays per lane and round, three lanes; extraction not included"
print(json.dumps(res))
