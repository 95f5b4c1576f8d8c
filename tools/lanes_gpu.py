"""The launcher's lane machinery on the real GPU, through the DROP-IN API (the reference scripts are not on the GPU box):
K host threads, each with its own HIP stream and its own pipeline object (UNet3DConditionModel + the re-bound guidance
functions, as t2v_video_sample.py:36-73 builds them), run `obtain_motion_representation` + `sample_video` on their share of
N synthetic examples at BASELINE config 2 (16 f x 512 x 512, 30 steps, 18 guided).  The DDIM steps replay from hipGraphs
captured per lane (thread-local capture).  Prints videos/min for lanes = 1 and lanes = K and whether every example's
latents are bit-identical between the two runs.

  python tools/lanes_gpu.py [--lanes 3] [--videos 6]"""
import argparse
import json
import sys
import threading
import time
import types

import torch

sys.path.insert(0, ".")
from motionclone_amd import lanes as mcl, lib, ops, spec  # noqa: E402
from motionclone_amd.engine import default_config  # noqa: E402
from motionclone_amd.models.unet import UNet3DConditionModel  # noqa: E402
from motionclone_amd.pipelines.pipeline_animation import AnimationPipeline  # noqa: E402
from motionclone_amd.scheduler import DDIMSchedulerState  # noqa: E402
from motionclone_amd.utils import motionclone_functions as mf  # noqa: E402

dev = torch.device("cuda:0")
N, G, GS = 30, 18, 0.4


def build_pipeline(sd):
    cfg = default_config()
    unet = UNet3DConditionModel(in_channels=4, out_channels=4, block_out_channels=cfg["block_out_channels"], layers_per_block=2,
                                cross_attention_dim=cfg["cross_attention_dim"], attention_head_dim=cfg["attention_heads"],
                                use_motion_module=True, motion_module_resolutions=[1, 2, 4, 8], motion_module_mid_block=False,
                                motion_module_kwargs=dict(num_attention_heads=cfg["motion_heads"], num_transformer_block=1,
                                                          attention_block_types=["Temporal_Self", "Temporal_Self"],
                                                          temporal_position_encoding=True))
    unet.load_state_dict(sd, strict=False)
    unet = unet.to(dev).to(dtype=torch.float16)
    pipe = AnimationPipeline(vae=None, text_encoder=None, tokenizer=None, unet=unet, controlnet=None,
                             scheduler=DDIMSchedulerState(beta_start=0.00085, beta_end=0.012, beta_schedule="linear",
                                                          steps_offset=1, clip_sample=False))
    pipe.scheduler.customized_step = mf.schedule_customized_step.__get__(pipe.scheduler)
    pipe.scheduler.customized_set_timesteps = mf.schedule_set_timesteps.__get__(pipe.scheduler)
    pipe.unet.forward = mf.unet_customized_forward.__get__(pipe.unet)
    for name in ("sample_video", "single_step_video", "get_temp_attn_prob", "add_noise", "compute_temp_loss",
                 "obtain_motion_representation"):
        setattr(pipe, name, getattr(mf, name).__get__(pipe))
    config = types.SimpleNamespace(cfg_scale=7.5, negative_prompt="", inference_steps=N, guidance_scale=GS, guidance_steps=G,
                                   warm_up_steps=10, cool_up_steps=10, motion_guidance_weight=2000,
                                   motion_guidance_blocks=["up_blocks.1"], add_noise_step=400, video_length=16, height=512,
                                   width=512, new_prompt="x")
    pipe.input_config, pipe.unet.input_config = config, config
    pipe.unet = mf.prep_unet_attention(pipe.unet, config.motion_guidance_blocks)
    pipe.unet = mf.prep_unet_conv(pipe.unet)
    pipe.scheduler.customized_set_timesteps(N, G, GS, device=dev, timestep_spacing_type="uneven")
    return pipe


def example(i):
    g = lambda s: torch.Generator(device=dev).manual_seed(s)   # noqa: E731
    text = torch.randn((2, 77, 768), generator=g(7 + i), device=dev).half()
    vid = (0.18215 * torch.randn((1, 4, 16, 64, 64), generator=g(11 + i), device=dev)).half()
    return text, vid, 2025 + i


def run(n_lanes, n_videos, sd, results, share):
    pipes = [build_pipeline(sd) for _ in range(n_lanes)]
    ops.set_gemm_share(share)     # the same tile / split-K choice in both runs: results must then be bit-identical
    errors = []

    def lane(k, timed):
        try:
            if n_lanes > 1:
                mcl.begin(k, n_lanes, dev)
                torch.cuda.set_stream(torch.cuda.Stream())
            for i in range(k, n_videos, n_lanes):
                text, vid, seed = example(i)
                gen = torch.Generator(device=dev).manual_seed(seed)
                pipes[k].obtain_motion_representation(generator=gen, motion_representation_path=None, video_latents=vid,
                                                      uncond_embeddings=text[0:1])
                gen = torch.Generator(device=dev).manual_seed(seed)
                out = pipes[k].sample_video(generator=gen, text_embeddings=text, decode=False)
                if timed:
                    results[(n_lanes, i)] = out.clone()
            torch.cuda.current_stream().synchronize()
        except BaseException as e:   # noqa: BLE001
            errors.append(e)
        finally:
            mcl.end()

    def sweep(timed):
        ts = [threading.Thread(target=lane, args=(k, timed)) for k in range(n_lanes)]
        for t in ts:
            t.start()
            if not timed:
                t.join()      # warm-up: one lane at a time (all hipGraph captures happen here, lanes.may_capture)
        for t in ts:
            t.join()
        if errors:
            raise errors[0]
        torch.cuda.synchronize()
    sweep(False)                       # warm-up: every lane captures its 30 step graphs
    t0 = time.perf_counter()
    sweep(True)
    dt = time.perf_counter() - t0
    ops.set_gemm_share(1)
    return 60.0 * n_videos / dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", type=int, default=3)
    ap.add_argument("--videos", type=int, default=6)
    a = ap.parse_args()
    lib.load()
    sd, _ = spec.synthetic_state_dict(default_config(), seed=1234, device=dev)
    sd = {k: v for k, v in sd.items()}
    results = {}
    r1 = run(1, a.videos, sd, results, a.lanes)
    rk = run(a.lanes, a.videos, sd, results, a.lanes)
    same = [bool(torch.equal(results[(1, i)], results[(a.lanes, i)])) for i in range(a.videos)]
    print(json.dumps(dict(videos=a.videos, lanes=a.lanes, videos_per_min_1_lane=r1, videos_per_min_lanes=rk,
                          speedup=rk / r1, bit_identical_to_one_lane=same,
                          note="drop-in API (obtain_motion_representation + sample_video, decode=False), config 2, hipGraph replay; "
                               "both runs with the GEMM share hint of the lane count (same kernels: results must be bit-identical)")))


if __name__ == "__main__":
    main()
