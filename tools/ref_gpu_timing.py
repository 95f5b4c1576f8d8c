#!/usr/bin/env python
"""The "before" GPU number of SURVEY.md 8(d): the reference's arithmetic (oracle/ restatement = same torch ops as the
reference's modules, pinned by tests/test_oracle_pins.py) run on the MI355X through stock PyTorch-ROCm, fp16 like
t2v_video_sample.py:19 (and fp32 for comparison): guided step, plain step and extraction timed separately, median of 3
after one warm-up, derived videos/min for the schedule of the configuration.

  python tools/ref_gpu_timing.py [--size 256|512] [--miopen 0|1] [--dtype f16|f32]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from motionclone_amd import spec  # noqa: E402
from motionclone_amd.engine import default_config  # noqa: E402
from oracle import guidance_ref as G  # noqa: E402


def med3(fn):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[1]


def run(size, miopen, dtype, sched):
    dev = torch.device("cuda:0")
    cfg = default_config()
    sd, _ = spec.synthetic_state_dict(cfg, seed=1234, device=dev)
    dt = torch.float16 if dtype == "f16" else torch.float32
    sd = {k: v.to(dt) for k, v in sd.items()}
    F, H = 16, size // 8
    g = lambda s: torch.Generator(device=dev).manual_seed(s)   # noqa: E731
    lat = torch.randn((1, 4, F, H, H), generator=g(2025), device=dev).to(dt)
    text = torch.randn((2, 77, 768), generator=g(7), device=dev).to(dt)
    vid = (0.18215 * torch.randn((1, 4, F, H, H), generator=g(11), device=dev)).to(dt)
    noise = torch.randn((1, 4, F, H, H), generator=g(2025), device=dev).to(dt)
    N, Gs, gs = sched
    ts = G.uneven_timesteps(N, Gs, gs)
    hp = dict(cfg_scale=7.5, motion_guidance_weight=2000.0, warm_up_steps=10, cool_up_steps=10, guidance_steps=Gs)
    out = {}
    with torch.backends.cudnn.flags(enabled=bool(miopen), benchmark=False):
        t0 = time.perf_counter()
        rep = G.extract_representation(sd, cfg, vid, noise, text[0:1])
        torch.cuda.synchronize()
        out["first_call_s"] = time.perf_counter() - t0      # includes any MIOpen / hipBLASLt first-use cost
        out["extraction_s"] = med3(lambda: G.extract_representation(sd, cfg, vid, noise, text[0:1]))
        out["guided_step_s"] = med3(lambda: G.guided_step(sd, cfg, lat, 0, ts, text, rep, hp))
        out["plain_step_s"] = med3(lambda: G.plain_step_full(sd, cfg, lat, Gs, ts, text, 7.5))
    sec = out["extraction_s"] + Gs * out["guided_step_s"] + (N - Gs) * out["plain_step_s"]
    out.update(sec_per_video=sec, videos_per_min=60.0 / sec, size=size, miopen=bool(miopen), dtype=dtype, schedule=sched,
               peak_mem_gib=torch.cuda.max_memory_allocated() / 2 ** 30, device=torch.cuda.get_device_name(0),
               torch=torch.__version__)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--miopen", type=int, default=1)
    ap.add_argument("--dtype", default="f16")
    a = ap.parse_args()
    sched = (10, 5, 0.3) if a.size == 256 else (30, 18, 0.4)
    print(json.dumps(run(a.size, a.miopen, a.dtype, sched)))
