"""Throughput with 1 vs 2 (vs 3) independent videos in flight on one GPU (separate HIP streams, one Python thread)."""
import json, sys, time, torch
sys.path.insert(0, ".")
from motionclone_amd import lib, spec
from motionclone_amd.engine import UNet3DEngine, default_config
from motionclone_amd.sampler import MotionCloneSampler
import bench
dev = torch.device("cuda:0")
lib.load()
cfg = default_config()
sd, _ = spec.synthetic_state_dict(cfg, seed=1234, device=dev)
eng = UNet3DEngine(sd, cfg, dev)
def mk(): return MotionCloneSampler(eng, num_inference_steps=30, guidance_steps=18, guidance_scale=0.4)
inputs = [bench.synth_inputs(dev, 16, 512, 512, s) for s in (42, 2026, 2025)]
def run(nstream, videos_per_stream=2, graphs=False):
    streams = [torch.cuda.Stream() for _ in range(nstream)]
    smps = [mk() for _ in range(nstream)]
    if graphs:
        for s in smps: s.enable_graphs()
    reps = [None] * nstream
    def prep(k):
        lat, text, vid, noise = inputs[k]
        with torch.cuda.stream(streams[k]):
            rep = smps[k].extract(vid, noise, text[0:1])
            reps[k] = smps[k].engine.prepare_representation(rep)
    def all_videos(nv):
        xs = [inputs[k][0] for k in range(nstream)]
        for v in range(nv):
            for k in range(nstream): prep(k)
            for i in range(30):
                for k in range(nstream):
                    with torch.cuda.stream(streams[k]):
                        xs[k] = smps[k].step(xs[k] if i else inputs[k][0], i, inputs[k][1], reps[k])
        return xs
    all_videos(1); torch.cuda.synchronize()
    t0 = time.perf_counter(); all_videos(videos_per_stream); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(streams=nstream, graphs=graphs, videos=videos_per_stream * nstream, seconds=dt, videos_per_min=60.0 * videos_per_stream * nstream / dt)
for ns, g in [(1, False), (2, False), (3, False), (1, True), (2, True)]:
    print(json.dumps(run(ns, graphs=g)), flush=True)
    torch.cuda.empty_cache()
