"""Experiment (round 4): V videos as ONE batch of 2V through the same launch sequence vs V videos on V streams.
Plain-step forward (B = 2 per video: uncond | cond) of the config-2 shape; reports ms per VIDEO-forward for
  streams x batch  in {1x2, 3x2, 1x4, 2x4, 1x6, 2x6, 1x8}
so that the two ways of keeping several examples in flight can be compared on the same kernels.
  python tools/batch_vs_streams.py [--size 512] [--reps 6] > gpurun_out/r04_batch_vs_streams.jsonl"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motionclone_amd import lib, ops, spec  # noqa: E402
from motionclone_amd.engine import UNet3DEngine, default_config  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--reps", type=int, default=6)
    args = ap.parse_args()
    lib.load()
    dev = torch.device("cuda", 0)
    cfg = default_config()
    sd, _ = spec.synthetic_state_dict(cfg, seed=1234, device=dev)
    eng = UNet3DEngine(sd, cfg, dev)
    H = args.size // 8
    g = torch.Generator(device=dev).manual_seed(1)
    for nstreams, B in [(1, 2), (3, 2), (1, 4), (2, 4), (1, 6), (2, 6), (1, 8), (1, 2)]:
        ops.set_gemm_share(nstreams)
        lat = [torch.randn((B, 4, args.frames, H, H), generator=g, device=dev, dtype=torch.float16) for _ in range(nstreams)]
        text = [torch.randn((B, 77, 768), generator=g, device=dev).half() for _ in range(nstreams)]
        streams = [torch.cuda.Stream() for _ in range(nstreams)]

        def run(n):
            for _ in range(n):
                for k in range(nstreams):
                    with torch.cuda.stream(streams[k]):
                        eng.forward(lat[k], 500, text[k])
        torch.cuda.synchronize()
        run(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.reps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        videos = nstreams * (B // 2) * args.reps
        print(json.dumps(dict(streams=nstreams, batch=B, ms_per_video_forward=1e3 * dt / videos,
                              peak_reserved_gib=torch.cuda.max_memory_reserved() / 2 ** 30)), flush=True)
        del lat, text
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
