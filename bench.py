#!/usr/bin/env python
"""Benchmark of the guided-DDIM hot path (BASELINE.json metric: videos/min + sec/denoise-step,
16f x 512x512 SD1.5 + AnimateDiff, MotionClone guidance).

A bench "step" is ONE VIDEO of BASELINE config 2: motion-representation extraction (partial UNet forward)
+ 30 DDIM steps of which the first 18 are guided (2 UNet forwards + guidance backward each) and 12 plain
(one B=2 forward), schedule (N, G, guidance_scale) = (30, 18, 0.4) as in SURVEY.md 8(d); VAE/CLIP are
outside the step metric and bypassed with synthetic tensors.  Synthetic random-init weights of the named
architecture and synthetic latents / text embeddings (there are no checkpoints offline).

Multi-GPU: replicas only - each rank samples its own (prompt, reference-video) pair; the only collective is
one RCCL broadcast of the packed fp16 weights from rank 0 before the timed region.

  python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from motionclone_amd.probe import LaunchProbe
from motionclone_amd import lib, ops, spec  # noqa: E402
from motionclone_amd.engine import UNet3DEngine, default_config  # noqa: E402
from motionclone_amd.sampler import MotionCloneSampler, sample_interleaved  # noqa: E402

# algorithmic work of the reference graph, FLOP = 2*MAC (measured on the reference's own code)
TFLOP_GUIDED, TFLOP_PLAIN, TFLOP_EXTRACT = 45.50, 35.35, 10.06
PEAK_FP16_MFMA_TFLOPS = 2500.0


def plan_rounds(nvideos, lanes):
    """round sizes (videos in flight together) for `nvideos` videos on up to `lanes` lanes: full rounds first, and a
    remainder of one is avoided by taking one video off the previous round (4 on 3 lanes -> [2, 2], 7 -> [3, 2, 2])"""
    rounds = []
    left = nvideos
    while left > 0:
        k = min(lanes, left)
        if left - k == 1 and k > 2:
            k -= 1
        rounds.append(k)
        left -= k
    return rounds


def synth_inputs(dev, F, H, W, seed):
    """latents = prepare_latents(seed) (pipeline_animation.py:316), text ~ N(0,1) [2,77,768],
    reference-video latents 0.18215 * N(0,1), extraction noise from the example seed."""
    g = torch.Generator(device=dev).manual_seed(seed)
    lat = torch.randn((1, 4, F, H // 8, W // 8), generator=g, device=dev, dtype=torch.float16)
    text = torch.randn((2, 77, 768), generator=torch.Generator(device=dev).manual_seed(7), device=dev).half()
    vid = (0.18215 * torch.randn((1, 4, F, H // 8, W // 8), generator=torch.Generator(device=dev).manual_seed(11),
                                 device=dev)).half()
    noise = torch.randn((1, 4, F, H // 8, W // 8), generator=torch.Generator(device=dev).manual_seed(seed),
                        device=dev, dtype=torch.float16)
    return lat, text, vid, noise


def one_video(smp, lat, text, vid, noise, step_events=None, ctrl=None):
    rep = smp.extract(vid, noise, text[0:1], add_noise_step=400, ctrl=ctrl)
    rep_dev = smp.engine.prepare_representation(rep)
    x = lat
    for i in range(len(smp.timesteps)):
        if step_events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        x = smp.step(x, i, text, rep_dev, ctrl=ctrl)
        if step_events is not None:
            e1.record()
            step_events.append((i < smp.G, e0, e1))
    return x


def vae_extras(dev, latents_tokens_or_lat, frames, size):
    """decode_latents (pipeline_animation.py:249-263) and vae.encode of the reference frames (motionclone_functions.py:64)
    on synthetic SD-1.5 AutoencoderKL weights, HIP path; seconds per video, median of 3."""
    from motionclone_amd.models.vae import vae_param_shapes
    from motionclone_amd.vae_engine import SD15_VAE_CONFIG, VaeDecoderEngine, VaeEncoderEngine
    g = torch.Generator(device=dev).manual_seed(4242)
    sd = {}
    for name, shape in vae_param_shapes(SD15_VAE_CONFIG).items():
        fan = 1
        for d in shape[1:]:
            fan *= d
        if "norm" in name:
            sd[name] = (1.0 if name.endswith("weight") else 0.0) + 0.1 * torch.randn(shape, generator=g, device=dev)
        elif name.endswith("bias"):
            sd[name] = 0.02 * torch.randn(shape, generator=g, device=dev)
        else:
            sd[name] = (torch.rand(shape, generator=g, device=dev) * 2 - 1) / fan ** 0.5
    dec = VaeDecoderEngine({k: v for k, v in sd.items() if k.startswith(("decoder.", "post_quant"))}, None, dev)
    enc = VaeEncoderEngine({k: v for k, v in sd.items() if k.startswith(("encoder.", "quant_conv"))}, None, dev)
    lat = (0.18215 * torch.randn((1, 4, frames, size // 8, size // 8), generator=g, device=dev)).half()
    img = (torch.rand((frames, 3, size, size), generator=g, device=dev) * 2 - 1).half()

    def med(fn):
        fn()
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[1]
    d = med(lambda: dec.decode_video(lat))
    e = med(lambda: enc.encode(img).sample())
    return dict(decode_sec_per_video=d, encode_sec_per_video=e, frames=frames, size=size,
                note="diffusers 0.16.0 AutoencoderKL architecture, synthetic weights; not part of `value`")


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(budget_s=75.0):
    """The oracle (CPU restatement of the reference path, oracle/unet3d_ref.py + guidance_ref.py, pinned to the unmodified
    reference by tests/test_oracle_pins.py) timed on this box's host cores, fp32, full SD1.5+AnimateDiff architecture, seeded
    synthetic weights, at BASELINE config 1's shape (16 f x 256 x 256, schedule (10, 5, 0.3)) as SURVEY.md 8(d) asks:
    extraction, plain step (one B=2 forward) and guided step (two forwards + autograd backward) are timed SEPARATELY after
    a warm-up forward, as many repetitions as a bounded time budget allows (median reported), and videos/min derived as
    extraction + 5 guided + 5 plain.  The FLOP-scaled config-2 figure is kept as a cross-check."""
    from oracle import guidance_ref as G
    from oracle import unet3d_ref as U
    cores = min(32, os.cpu_count() or 1)   # more threads oversubscribe the small per-frame ops (measured: >10x slower)
    torch.set_num_threads(cores)
    cfg = U.SD15_CONFIG
    sd = U.random_state_dict(cfg, seed=1234)
    F, H = 16, 32
    g = lambda s: torch.Generator().manual_seed(s)   # noqa: E731
    lat = torch.randn(1, 4, F, H, H, generator=g(2025))
    text = torch.randn(2, 77, 768, generator=g(7))
    vid = 0.18215 * torch.randn(1, 4, F, H, H, generator=g(11))
    noise = torch.randn(1, 4, F, H, H, generator=g(2025))
    ts = G.uneven_timesteps(10, 5, 0.3)
    hp = dict(cfg_scale=7.5, motion_guidance_weight=2000.0, warm_up_steps=10, cool_up_steps=10, guidance_steps=5)
    t_start = time.time()
    with torch.no_grad():
        U.unet_forward(sd, cfg, lat[:, :, :2, :8, :8].contiguous(), 500, text[:1])   # page in / thread-pool warm-up

    def timed(fn, max_reps=3):
        ts_ = []
        for _ in range(max_reps):
            t0 = time.time()
            out = fn()
            ts_.append(time.time() - t0)
            if time.time() - t_start > budget_s:
                break
        return sorted(ts_)[len(ts_) // 2], len(ts_), out
    t_ext, n_ext, rep = timed(lambda: G.extract_representation(sd, cfg, vid, noise, text[0:1]))
    t_plain, n_plain, _ = timed(lambda: G.plain_step_full(sd, cfg, lat, 5, ts, text, 7.5))
    t_guided, n_guided, _ = timed(lambda: G.guided_step(sd, cfg, lat, 0, ts, text, rep, hp))
    sec_cfg1 = t_ext + 5 * t_guided + 5 * t_plain
    tflop_cfg1 = 2.39 + 5 * 10.41 + 5 * 8.17
    tflop_cfg2 = 18 * TFLOP_GUIDED + 12 * TFLOP_PLAIN + TFLOP_EXTRACT
    return dict(value=60.0 / sec_cfg1, unit="videos/min (BASELINE config 1: 16f x 256x256, schedule (10,5,0.3), UNet only)",
                cores=cores, cpu_model=_cpu_model(), kind="port",
                sample="oracle fp32 on %d threads, 16f x 256x256: extraction %.2f s (median of %d), plain step %.2f s (%d), "
                       "guided step %.2f s (%d); one video = extraction + 5 guided + 5 plain = %.1f s"
                       % (cores, t_ext, n_ext, t_plain, n_plain, t_guided, n_guided, sec_cfg1),
                extraction_s=t_ext, plain_step_s=t_plain, guided_step_s=t_guided, sec_per_video_config1=sec_cfg1,
                tflops=tflop_cfg1 / sec_cfg1,
                config2_videos_per_min_flop_scaled=60.0 / (sec_cfg1 * tflop_cfg2 / tflop_cfg1))


def reference_gpu_baseline(dev, size=512, sched=(30, 18, 0.4)):
    """SURVEY.md 8(d) "before" number: the reference's arithmetic (the oracle = the same torch ops as the reference's modules)
    on this MI355X through stock PyTorch-ROCm in fp16 (t2v_video_sample.py:19), without xformers (unavailable on ROCm):
    guided step, plain step and extraction, median of 3 after a warm-up.  Convolutions use PyTorch's own GPU path (MIOpen
    off: a fresh box has no precompiled gfx950 MIOpen kernels and would JIT every shape)."""
    from oracle import guidance_ref as G
    cfg = default_config()
    sd, _ = spec.synthetic_state_dict(cfg, seed=1234, device=dev)
    F, H = 16, size // 8
    g = lambda s: torch.Generator(device=dev).manual_seed(s)   # noqa: E731
    lat = torch.randn((1, 4, F, H, H), generator=g(2025), device=dev).half()
    text = torch.randn((2, 77, 768), generator=g(7), device=dev).half()
    vid = (0.18215 * torch.randn((1, 4, F, H, H), generator=g(11), device=dev)).half()
    noise = torch.randn((1, 4, F, H, H), generator=g(2025), device=dev).half()
    N, Gs, gs = sched
    ts = G.uneven_timesteps(N, Gs, gs)
    hp = dict(cfg_scale=7.5, motion_guidance_weight=2000.0, warm_up_steps=10, cool_up_steps=10, guidance_steps=Gs)

    def med3(fn):
        fn()
        torch.cuda.synchronize()
        ts_ = []
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts_.append(time.perf_counter() - t0)
        return sorted(ts_)[1]
    with torch.backends.cudnn.flags(enabled=False, benchmark=False):
        rep = G.extract_representation(sd, cfg, vid, noise, text[0:1])
        t_ext = med3(lambda: G.extract_representation(sd, cfg, vid, noise, text[0:1]))
        t_guided = med3(lambda: G.guided_step(sd, cfg, lat, 0, ts, text, rep, hp))
        t_plain = med3(lambda: G.plain_step_full(sd, cfg, lat, Gs, ts, text, 7.5))
    sec = t_ext + Gs * t_guided + (N - Gs) * t_plain
    torch.cuda.empty_cache()
    return dict(videos_per_min=60.0 / sec, sec_per_video=sec, extraction_s=t_ext, guided_step_s=t_guided, plain_step_s=t_plain,
                dtype="f16", note="oracle (reference arithmetic) via stock PyTorch %s ROCm on %s, MIOpen off, no xformers; "
                                  "config-2 shape, schedule %s" % (torch.__version__, torch.cuda.get_device_name(0), list(sched)))


TIMED_DURATIONS_FILE = os.path.join("profiles", "kernel_durations_timed.json")
_G5_TILES = {(256, 320): "gemm5<256x320>", (128, 320): "gemm5<128x320>", (256, 160): "gemm5<256x160 x2 per CU>",
             (256, 256): "gemm5<256x256, 4 waves>"}
_MODE_NAMES = ["DENSE", "CONV_S1", "CONV_S2", "CONV_UP", "TCONV_S2"]


def family_of_traced_kernel(name):
    """rocprofv3 kernel name -> the probe's family name (motionclone_amd/probe.py) for the GEMM structures; None otherwise.
    gemm5_kernel<MODE, EPI, VAR, BM, BN, waves, stages> (EPI 1 = fused GEGLU and split-K launches are the same family as in
    the probe, which names a GEMM by structure and mode only); gemm4_kernel<KS, GEGLU, NORM>."""
    import re
    m = re.search(r"gemm5_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)", name)
    if m:
        mode, bm, bn = int(m.group(1)), int(m.group(4)), int(m.group(5))
        base = _G5_TILES.get((bm, bn))
        return "%s %s" % (base, _MODE_NAMES[mode]) if base and mode < len(_MODE_NAMES) else None
    m = re.search(r"gemm6_kernel<(\d+), (\d+), (\d+), (\d+)>", name)      # <EPI, RES, VAR, stream-K>
    if m:
        return "gemm6<256x320 tile loop%s> DENSE" % (", stream-K" if int(m.group(4)) else "")
    m = re.search(r"gemm4_kernel<(\d+), (\w+), (\d+)>", name)
    if m:
        return "gemm4<K=320 streaming> " + {0: "DENSE", 1: "LayerNorm + DENSE", 2: "GroupNorm + DENSE"}.get(int(m.group(3)), "?")
    return None


def loaded_lib_stamp():
    """Source stamp of the HIP library this process loaded (written by motionclone_amd/build.py next to the .so); None if absent."""
    from motionclone_amd import lib
    try:
        return open(lib.HIP_LIB_PATH + ".stamp").read().strip()
    except OSError:
        return None


def timed_roofline(roof_all, prof, lanes=3, batch=1):
    """The dominant GEMM family IN THE TIMED REGIME (hipGraph replay, several videos in flight): HIP events cannot bracket
    launches inside a graph replay, so the per-launch DURATION comes from a rocprofv3 --kernel-trace --stats run of this same
    command (profiles/kernel_durations_timed.json, written by tools/kernel_stats_md.py from the round's profile run; its
    `code` field names the commit) and the algorithmic FLOP per launch from this run's probe video, which issues the same
    launch sequence with the same tile choice.  `overlap` = summed kernel time / wall time of the traced bench region: kernels
    of the videos in flight share the CUs, so a kernel's own duration is longer than alone; achieved x overlap is the rate the
    family's launches sustain per unit of GPU time they occupy."""
    if not prof or not roof_all:
        return None
    # a trace is only this run's kernels if it was taken from THIS library on THIS device: tools/kernel_stats_md.py records the
    # library's source stamp (build.py's sha1 of sources + flags, next to the .so) and the device name; anything else is refused
    stamp, devname = loaded_lib_stamp(), (torch.cuda.get_device_name(0) if torch.cuda.is_available() else None)
    if (prof.get("lanes", 3), prof.get("batch", 1)) != (lanes, batch):
        return dict(achieved=None, frac=None, source=TIMED_DURATIONS_FILE, code=prof.get("code"),
                    reason="the trace was taken with %s lanes x %s videos, this run has %d x %d" % (prof.get("lanes", 3), prof.get("batch", 1), lanes, batch))
    if prof.get("lib_stamp") is None or prof.get("lib_stamp") != stamp or prof.get("device") != devname:
        return dict(achieved=None, frac=None, source=TIMED_DURATIONS_FILE, code=prof.get("code"),
                    reason="stale trace: recorded for library stamp %s on %s, this run loaded %s on %s - re-run the profile (tools/gpu_profile.sh)"
                           % (str(prof.get("lib_stamp"))[:12], prof.get("device"), str(stamp)[:12], devname))
    groups = {}
    for name, k in prof.get("kernels", {}).items():
        fam = family_of_traced_kernel(name)
        if fam is None:
            continue
        g = groups.setdefault(fam, dict(calls=0, total_us=0.0))
        g["calls"] += k["calls"]
        g["total_us"] += k["calls"] * k["avg_us"]
    best = None
    for fam, g in groups.items():
        rows = [r for n, r in roof_all.items() if n == fam or n == fam + " split-K + reduce"]
        if not rows or not g["calls"]:
            continue
        flop = sum(r["flop_per_launch"] * r["launches"] for r in rows) / sum(r["launches"] for r in rows)
        if flop <= 0:
            continue
        avg_us = g["total_us"] / g["calls"]
        tf = flop / avg_us / 1e6
        row = dict(kernel=fam, bound="mfma", achieved=tf, peak=PEAK_FP16_MFMA_TFLOPS, unit="TFLOP/s", frac=tf / PEAK_FP16_MFMA_TFLOPS,
                   avg_launch_us=avg_us, flop_per_launch=flop, share_of_kernel_time=g["total_us"] / 1e6 / prof["total_kernel_s"],
                   overlap=prof.get("overlap"), regime=prof.get("regime"), source=TIMED_DURATIONS_FILE, code=prof.get("code"))
        if best is None or row["share_of_kernel_time"] > best["share_of_kernel_time"]:
            best = row
    return best


def auto_packing(steps, frames, size, sparsectrl):
    """(lanes, videos batched per lane) when --inflight / --batch are not given.  Independent videos are the unit of parallelism
    (SURVEY.md 8e); a GPU holds several: `lanes` launch sequences on their own streams (kernel tails and the small 16x16 / 8x8-level
    launches of one are filled by the others) x `batch` videos that go through ONE launch sequence ([V, ...] latents: the same
    kernels on V times the rows - fewer, larger launches, the small levels fill the chip).  Measured at config 2 on one MI355X
    (profiles/r06_packing.txt, videos/min, GiB reserved): 3 x 1 36.8 (46), 3 x 2 38.0 (83), 2 x 3 37.7 (83), 2 x 4 38.3 (106),
    2 x 5 38.8 (131), 2 x 6 39.0 (157), 3 x 4 39.1 (156), 1 x 8 37.8 (110).  The choice: two lanes, up to five videos each
    (131 GiB of the 288), with lanes x batch dividing --steps so that every round is a full one; other shapes keep round 5's
    three lanes x one video (config 5's 32 f x 768^2 holds 2 lanes; SparseCtrl is not batched)."""
    if sparsectrl or (frames, size) != (16, 512):
        return (2 if frames * size * size > 16 * 512 * 512 else 3), 1
    for vb in (5, 4, 3):
        if steps % (2 * vb) == 0:
            return 2, vb
    for nf, vb in ((3, 2), (2, 2)):
        if steps % (nf * vb) == 0:
            return nf, vb
    return 3, 1


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: replace this process by `torch.distributed.run` with N ranks on this
    node (one per GPU, rendezvous on 127.0.0.1 and a free port) running the same command line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def compact_line(res, detail_path, limit=4000):
    """The ONE line the driver parses: the contract keys, the dominant kernel's roofline row and the CPU baseline, nothing
    long.  Everything else (`roofline_by_kernel`, traffic per shape, VAE, eager loop, baselines) goes to `detail_path`."""
    keep = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "sec_per_guided_step", "sec_per_plain_step", "sec_per_denoise_step",
            "sec_per_denoise_step_throughput", "e2e_tflops_per_gpu", "e2e_frac_of_mfma_peak", "videos_per_min_incl_vae"]
    line = {k: res[k] for k in keep if k in res}
    r = res.get("roofline")
    if r:
        line["roofline"] = {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                                                  "traffic_vs_algorithmic", "traffic_launch_coverage", "avg_launch_us", "launches",
                                                  "share_of_probe_video", "regime")}
    rt = res.get("roofline_timed")
    if rt:
        line["roofline_timed"] = {k: rt.get(k) for k in ("kernel", "achieved", "frac", "avg_launch_us", "overlap",
                                                         "share_of_kernel_time", "source", "code", "reason")
                                  if rt.get(k) is not None or k in ("achieved", "frac")}
    c = res.get("cpu_baseline")
    if c:
        line["cpu_baseline"] = {k: c.get(k) for k in ("value", "unit", "cores", "cpu_model", "kind", "sample", "error")
                                if c.get(k) is not None}
    g = res.get("reference_gpu_baseline")
    if g and "videos_per_min" in g:
        line["stock_pytorch_gpu_videos_per_min"] = g["videos_per_min"]
    if res.get("eager"):
        line["eager_one_video_at_a_time_videos_per_min"] = res["eager"]["videos_per_min"]
        line["identical_to_eager_path"] = res["eager"]["identical_to_graph_path"]
    if res.get("three_lanes_x_one_video") and "videos_per_min" in res["three_lanes_x_one_video"]:
        line["three_lanes_x_one_video_videos_per_min"] = res["three_lanes_x_one_video"]["videos_per_min"]
    if res.get("hbm_footprint"):
        line["peak_reserved_gib"] = res["hbm_footprint"]["peak_reserved_gib"]
    if "warmup_seconds" in res:
        line["cold_start_seconds"] = res["warmup_seconds"]     # warm-up videos incl. the graph captures of every lane
    line["detail"] = detail_path
    # never let a long string cost the record (the whole benchmark has run by now): shed, in this order, the prose, the
    # error texts, the optional objects, and finally cut the strings of the contract keys - a line is ALWAYS printed
    contract = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better")

    def shed(stage):
        cb, cfg = line.get("cpu_baseline"), line.get("config", {})
        if stage == 0 and cb:
            cb.pop("sample", None)
        elif stage == 1 and cb and "error" in cb:
            cb["error"] = str(cb["error"])[:120]
        elif stage == 2:
            line.pop("roofline_timed", None)
        elif stage == 3:
            line.pop("cpu_baseline", None)
        elif stage == 4:
            line.pop("roofline", None)
        elif stage == 5:
            for k in [k for k in line if k not in keep and k != "detail"]:
                line.pop(k)
        elif stage == 6:
            cfg["workload"] = str(cfg.get("workload", ""))[:160]
        elif stage == 7:
            line["metric"] = str(line.get("metric", ""))[:160]
        elif stage == 8:
            line["config"] = {"workload": str(cfg.get("workload", ""))[:80]}
        elif stage == 9:
            for k in [k for k in line if k not in contract]:
                line.pop(k)
    out = json.dumps(line)
    for stage in range(10):
        if len(out) <= limit:
            break
        shed(stage)
        out = json.dumps(line)
    return out


def write_detail(res):
    """full record -> profiles/r06_bench_detail.json (and gpurun_out/, which is what travels back from a GPU box)"""
    rel = os.path.join("profiles", "r06_bench_detail.json")
    for d in ("profiles", "gpurun_out"):
        try:
            os.makedirs(os.path.join(ROOT, d), exist_ok=True)
            with open(os.path.join(ROOT, d, "r06_bench_detail.json"), "w") as f:
                json.dump(res, f, indent=1)
        except OSError:
            pass
    return rel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="videos timed per GPU (10 = one round of the automatic packing, 2 lanes x 5)")
    ap.add_argument("--warmup", type=int, default=1, help="untimed warm-up videos per GPU")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ddim-steps", type=int, default=30)
    ap.add_argument("--guided-steps", type=int, default=18)
    ap.add_argument("--guidance-scale", type=float, default=0.4)
    ap.add_argument("--sparsectrl", action="store_true", help="BASELINE config 4: add the SparseCtrl (i2v_rgb) encoder pass")
    ap.add_argument("--no-graphs", action="store_true", help="time the eager launch sequence instead of the per-step hipGraphs")
    ap.add_argument("--gemm-lanes", type=int, default=0, help="value handed to ops.set_gemm_share (0 = the number of videos in flight)")
    ap.add_argument("--shapes-out", default=None, help="write the per-(kernel, shape) GEMM time table of the probe video to this JSON file")
    ap.add_argument("--no-gn-epilogue", action="store_true", help="A/B: GroupNorm statistics always from their own pass over the tensor "
                    "(round 5) instead of the producing GEMM's epilogue (mc_gemm_gnstats_f16)")
    ap.add_argument("--no-norm-fusion", action="store_true", help="A/B: LayerNorm / GroupNorm as separate launches in front of the "
                    "K = 320 GEMMs (the round-3 launch sequence) instead of mc_norm_gemm_f16")
    ap.add_argument("--no-shared-prefix", action="store_true", help="A/B: feed the UNet the duplicated CFG batch [x | x] as rounds 1-4 did, "
                    "instead of running what precedes the first cross-attention once for both halves (engine.forward: dup)")
    ap.add_argument("--two-b1-guided", action="store_true", help="A/B: guided steps as the reference issues them - eps_u and eps_c as two "
                    "B = 1 forwards (the un-guided one keeps nothing for a backward) instead of one B = 2 forward")
    ap.add_argument("--no-probe", action="store_true", help="skip the eager / roofline-probe videos after the timed region (profiling "
                    "runs that should hold the timed regime's kernels only: tools/gpu_profile_r05.sh)")
    ap.add_argument("--probe-one-lane", action="store_true", help="roofline probe video with the tile choice of ONE video in flight "
                    "(round 4's probe regime) instead of the timed region's")
    ap.add_argument("--no-detail", action="store_true", help="do not write profiles/r06_bench_detail.json")
    ap.add_argument("--no-vae", action="store_true", help="skip the (untimed, informational) VAE decode / encode measurement")
    ap.add_argument("--batch", type=int, default=0, help="(0 = automatic, see auto_packing) videos batched into ONE launch sequence per lane (latents [V, ...], text "
                    "[u_1 .. u_V | c_1 .. c_V]: the same kernels on V times the rows); --inflight lanes x --batch videos are in "
                    "flight together.  --steps must be a multiple of it")
    ap.add_argument("--tileloop", choices=["auto", "off", "all"], default="auto", help="A/B: the persistent tile loop (gemm6.hip) for the "
                    "dense 256x320-tile layers: auto = the library's measured policy, off = never (round 5's kernels), all = every "
                    "shape the kernel accepts")
    ap.add_argument("--inflight", type=int, default=0, help="(0 = automatic, see auto_packing) independent videos processed concurrently per GPU (own HIP stream, "
                    "own sampler / graphs each).  At config 2: 2 in flight +8-10 %% videos/min over one (kernel tails and the "
                    "small 16x16 / 8x8-level kernels of one video are filled by the others), 3 in flight another +2.6 %%, 4 lose; "
                    "results bit-identical to the one-at-a-time run (checked in the run: `eager.identical_to_graph_path`, "
                    "tools/concurrency_check.py).  --steps videos are processed in rounds of up to this many, never leaving "
                    "a single video for the last round when it can be avoided (4 -> 2 + 2, 5 -> 3 + 2, 7 -> 3 + 2 + 2)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)     # plain `python bench.py --gpus N`: become the N-rank launcher (does not return)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, "
                         "or without a launcher: bench.py starts its own ranks)" % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # rank 0 alone runs the eager / probe videos and the baselines after the timed region while the others wait in the final
        # barrier: a timeout well past that post-work, so RCCL's watchdog never tears the job down before the line is printed
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(hours=2))
    lib.load()  # fails loudly if the gfx950 library is missing
    if args.no_gn_epilogue:
        ops.GN_FROM_EPILOGUE = False
    if args.no_norm_fusion:
        ops.NORM_GEMM_MIN_ROWS = 1 << 62
    ops.TILELOOP = {"auto": None, "off": False, "all": True}[args.tileloop]

    cfg = default_config()
    total = sum(int(torch.Size(s).numel()) for s in spec.param_shapes(cfg).values())
    flat = torch.empty(total, dtype=torch.float16, device=dev)
    if rank == 0:
        sd, flat = spec.synthetic_state_dict(cfg, seed=1234, device=dev, flat=flat)
    if world > 1:
        dist.broadcast(flat, src=0)  # RCCL over xGMI: 2.38 GiB once, outside the timed region
    if rank != 0:
        sd, off = {}, 0
        for name, shape in spec.param_shapes(cfg).items():
            n = int(torch.Size(shape).numel())
            sd[name] = flat[off:off + n].view(shape)
            off += n
    eng = UNet3DEngine(sd, cfg, dev)
    eng.share_prefix = not args.no_shared_prefix
    N_STEPS, G_STEPS, G_SCALE = args.ddim_steps, args.guided_steps, args.guidance_scale
    ceng = None
    if args.sparsectrl:
        from motionclone_amd.engine import ControlNetEngine
        ceng = ControlNetEngine(spec.synthetic_controlnet_state_dict(cfg, seed=4321, device=dev), cfg, dev)
        ceng.share_prefix = eng.share_prefix     # --no-shared-prefix is an A/B of BOTH networks
    smp = MotionCloneSampler(eng, cfg_scale=7.5, motion_guidance_weight=2000.0, warm_up_steps=10, cool_up_steps=10,
                             num_inference_steps=N_STEPS, guidance_steps=G_STEPS, guidance_scale=G_SCALE, controlnet=ceng,
                             batch_guided=not args.two_b1_guided)
    # per-rank example seeds as in configs/t2v_camera.jsonl (42, 42, 2026, default 2025, ...)
    seeds = [42, 42, 2026, 2025, 2026, 2026, 2026, 2025]
    lat, text, vid, noise = synth_inputs(dev, args.frames, args.size, args.size, seeds[rank % len(seeds)])
    ctrl = None
    if args.sparsectrl:   # one condition image (its VAE latent, synthetic) on frame 0, as configs/i2v_rgb.jsonl does
        cond = torch.zeros_like(vid)
        mask = torch.zeros_like(vid[:, :1])
        cond[:, :, 0] = vid[:, :, 0]
        mask[:, :, 0] = 1
        ctrl = dict(cond=cond, mask=mask, scale=1.0)

    # The timed path replays one hipGraph per DDIM step (sampler.enable_graphs: bit-identical to the eager launches,
    # tests/test_fullsize_properties.py); everything that differs between videos enters through static buffers.  The
    # motion-representation extraction (once per video) stays eager.  `--no-graphs` times the eager launch sequence instead.
    probe = LaunchProbe().install()
    use_graphs = not args.no_graphs
    # Independent (prompt, reference-video) samples are the unit of parallelism of this workload.  `--inflight`
    # of them can run concurrently inside one GPU, each on its own HIP stream with its own sampler (and graphs): the launch
    # sequence of one video leaves CUs idle in kernel tails and in the small 16x16 / 8x8-level kernels, which a second video
    # fills (measured +8-10 % videos/min for 2 in flight, nothing more for 3: tools/concurrency_probe.py).  Results are
    # bit-identical to the one-at-a-time run since the library is built without packed-fp32 VALU code (csrc/temporal.hip).
    auto_nf, auto_vb = auto_packing(args.steps, args.frames, args.size, args.sparsectrl)
    if args.batch <= 0:      # a given --inflight alone keeps round 5's one video per lane
        args.batch = auto_vb if args.inflight <= 0 else 1
    if args.inflight <= 0:
        args.inflight = auto_nf if args.batch == auto_vb else 3
    VB = max(1, args.batch)
    if args.steps % VB or (args.sparsectrl and VB > 1):
        raise SystemExit("bench.py: --steps must be a multiple of --batch (and --batch 1 with --sparsectrl)")
    NF = max(1, min(args.inflight, args.steps // VB))
    ops.set_gemm_share(args.gemm_lanes or NF)      # every launch of this process, timed region and probe alike (tile / split-K choice only)
    streams = [torch.cuda.Stream(device=dev) for _ in range(NF)]
    smps = [smp] + [MotionCloneSampler(eng, cfg_scale=7.5, motion_guidance_weight=2000.0, warm_up_steps=10, cool_up_steps=10,
                                       num_inference_steps=N_STEPS, guidance_steps=G_STEPS, guidance_scale=G_SCALE,
                                       controlnet=ceng, batch_guided=not args.two_b1_guided) for _ in range(NF - 1)]
    if use_graphs:
        for sm in smps:
            sm.enable_graphs()
    # every video of the in-flight set is its own example (different seeds = different latents / noise); a lane's job is one
    # video, or a list of --batch videos that go through one launch sequence
    all_inputs = [(lat, text, vid, noise)] + [synth_inputs(dev, args.frames, args.size, args.size, seeds[(rank + 1 + k) % len(seeds)] + 7 * (k + 1))
                                               for k in range(NF * VB - 1)]
    lane_inputs = [all_inputs[k] if VB == 1 else all_inputs[k * VB:(k + 1) * VB] for k in range(NF)]

    def run_videos(nvideos, step_events=None):
        """nvideos videos, NF lanes x VB videos at a time (motionclone_amd.sampler.sample_interleaved)"""
        last = None
        pending = {}

        def on_step(k, i, enter):
            if step_events is None:
                return
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            if enter:
                pending[k] = ev
            else:
                step_events.append((i < G_STEPS, pending.pop(k), ev))

        for k_act in plan_rounds(nvideos // VB, NF):
            xs = sample_interleaved(smps[:k_act], lane_inputs[:k_act], streams[:k_act], add_noise_step=400, ctrl=ctrl,
                                    on_step=on_step)
            last = xs[0]
        return last

    warm_videos = max(NF * VB, (args.warmup + VB - 1) // VB * VB)     # every lane's first pass captures its graphs / fills its allocator pool: never timed
    t_warm0 = time.perf_counter()
    run_videos(warm_videos)
    torch.cuda.synchronize()
    warm_seconds = time.perf_counter() - t_warm0      # cold start: lazy weight packing, one eager pass per kind of step, 30 captures per lane
    # what the warm-up's eager passes left cached in the ordinary pool is of no use to the replays (they live in the lanes'
    # graph pools): hand it back, and count the footprint of the timed region on its own
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    step_events = []
    t0 = time.perf_counter()
    out = run_videos(args.steps, step_events)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timed_reserved = torch.cuda.max_memory_reserved(dev) / 2 ** 30
    timed_allocated = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    assert torch.isfinite(out.float()).all(), "non-finite latents"
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # Roofline probe: ONE more video on the eager launch sequence (same kernels, same order as the graphs hold), every GEMM /
    # conv launch bracketed by HIP events on the launch stream.  Outside the timed region: events cannot bracket launches
    # inside a graph replay.  `eager` also reports what the un-graphed loop costs on this box.
    eager_info = None
    probe_elapsed = None
    if rank == 0 and not args.no_probe:
        sme = MotionCloneSampler(eng, cfg_scale=7.5, motion_guidance_weight=2000.0, warm_up_steps=10, cool_up_steps=10,
                                 num_inference_steps=N_STEPS, guidance_steps=G_STEPS, guidance_scale=G_SCALE, controlnet=ceng,
                                 batch_guided=not args.two_b1_guided)
        def eager_lane0():     # lane 0's job (one video, or its --batch videos as one launch sequence) without graphs
            if VB == 1:
                return one_video(sme, lat, text, vid, noise, ctrl=ctrl)
            return sample_interleaved([sme], [lane_inputs[0]], None, add_noise_step=400, ctrl=ctrl)[0]
        out_e = eager_lane0()      # untimed eager warm-up
        torch.cuda.synchronize()
        te0 = time.perf_counter()
        out_e = eager_lane0()
        torch.cuda.synchronize()
        te = (time.perf_counter() - te0) / VB
        # Round 5: the probe video runs with the tile / split-K choice of the TIMED region (the kernels `value` is made of; round 4
        # probed the one-video-in-flight choice, which moved the wide layers to another kernel and hid their HBM re-reads from
        # the `roofline` row).  --probe-one-lane restores the old regime for tools.
        if args.probe_one_lane:
            ops.set_gemm_share(1)
            one_video(sme, lat, text, vid, noise, ctrl=ctrl)          # the single-lane choice may touch new kernels: warm
            torch.cuda.synchronize()
        probe.enabled = True
        tp0 = time.perf_counter()
        if args.probe_one_lane:
            one_video(sme, lat, text, vid, noise, ctrl=ctrl)
        else:
            eager_lane0()          # lane 0's job as the timed region runs it: its --batch videos as ONE launch sequence
        torch.cuda.synchronize()
        probe_elapsed = time.perf_counter() - tp0
        probe.enabled = False
        ops.set_gemm_share(args.gemm_lanes or NF)
        eager_info = dict(videos_per_min=60.0 / te, sec_per_video=te, identical_to_graph_path=bool(torch.equal(out_e, out)),
                          note="same launch sequence without hipGraphs, one lane (its --batch videos); not part of `value`")
        del sme
    # Round 6: next to the packed `value`, the regime of rounds 3 - 5 - THREE lanes x ONE video, which is also what the launcher
    # (motionclone_amd.launch --lanes 3) gives the unmodified entry scripts, whose calls carry one example each.  Six videos
    # after their own warm-up round, outside the timed region.
    lanes_only = None
    if rank == 0 and world == 1 and not args.no_probe and use_graphs and VB > 1 and not args.sparsectrl:
        try:
            torch.cuda.empty_cache()       # what the eager / probe passes left cached (~60 GiB at 5 batched videos)
            ops.set_gemm_share(3)
            sm3 = [MotionCloneSampler(eng, cfg_scale=7.5, motion_guidance_weight=2000.0, warm_up_steps=10, cool_up_steps=10,
                                      num_inference_steps=N_STEPS, guidance_steps=G_STEPS, guidance_scale=G_SCALE) for _ in range(3)]
            for sm in sm3:
                sm.enable_graphs()
            st3 = [torch.cuda.Stream(device=dev) for _ in range(3)]
            sample_interleaved(sm3, all_inputs[:3], st3, add_noise_step=400)       # captures
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            for _ in range(2):
                sample_interleaved(sm3, all_inputs[:3], st3, add_noise_step=400)
            torch.cuda.synchronize()
            lanes_only = dict(videos_per_min=6 * 60.0 / (time.perf_counter() - t3), lanes=3, videos_per_lane=1, videos=6,
                              note="round 5's regime (one video per launch sequence): what `motionclone_amd.launch --lanes 3` runs "
                                   "for the unmodified entry scripts; `value` batches %d videos per lane through the sampler API" % VB)
            del sm3
        except Exception as e:   # noqa: BLE001  (an extra: never costs the record)
            lanes_only = {"error": "%s: %s" % (type(e).__name__, e)}
        ops.set_gemm_share(args.gemm_lanes or NF)
    graph_info = dict(enabled=use_graphs, graphs=sum(len(sm._graphs) for sm in smps) if use_graphs else 0,
                      note="one graph per DDIM step, captured during warm-up; latents / text / representation refreshed by "
                           "device copies into static buffers before each replay; extraction eager")
    vae_info = None
    if rank == 0 and world == 1 and not args.no_vae:
        vae_info = vae_extras(dev, out, args.frames, args.size)
    if rank == 0:
        gsec = [e0.elapsed_time(e1) / 1e3 for g, e0, e1 in step_events if g]
        psec = [e0.elapsed_time(e1) / 1e3 for g, e0, e1 in step_events if not g]
        videos = args.steps * world
        # algorithmic work of the reference graph per step; only tabulated for the BASELINE shapes
        table = {(16, 256): (10.41, 8.17, 2.39), (16, 512): (45.50, 35.35, 10.06), (32, 768): (235.9, 181.1, 49.7)}
        tg, tp, te = table.get((args.frames, args.size), (float("nan"),) * 3)
        tflop_video = G_STEPS * tg + (N_STEPS - G_STEPS) * tp + te
        roof_all = probe.summary(probe_elapsed) if probe_elapsed else {}
        by_shape = probe.by_shape() if probe_elapsed else []
        if args.shapes_out:
            json.dump(by_shape, open(args.shapes_out, "w"), indent=1)
        # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md), per SHAPE, next to
        # that shape's algorithmic bytes; a family's `traffic` is the launch-weighted mean over its measured shapes and
        # `traffic_vs_algorithmic` the ratio on exactly those shapes
        traffic_rows = []
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic_per_shape.json")
        if os.path.exists(tfile) and (args.frames, args.size) == (16, 512):
            pmc = json.load(open(tfile))
            for row in by_shape:
                for m in pmc:
                    if m["kernel"] == row["kernel"] and list(m["shape"]) == list(row["shape"]):
                        traffic_rows.append(dict(kernel=row["kernel"], shape=row["shape"], launches=row["launches"],
                                                 traffic_bytes_per_launch=m["traffic_bytes"],
                                                 algorithmic_bytes_per_launch=row["algorithmic_bytes_per_launch"],
                                                 ratio=m["traffic_bytes"] / row["algorithmic_bytes_per_launch"]))
            for fam, r in roof_all.items():
                sel = [t for t in traffic_rows if t["kernel"] == fam]
                if sel:
                    n = sum(t["launches"] for t in sel)
                    r["traffic"] = sum(t["launches"] * t["traffic_bytes_per_launch"] for t in sel) / n
                    r["traffic_vs_algorithmic"] = r["traffic"] / (sum(t["launches"] * t["algorithmic_bytes_per_launch"] for t in sel) / n)
                    r["traffic_launch_coverage"] = n / r["launches"]
        roof = None
        if roof_all:
            dom = max(roof_all, key=lambda n: roof_all[n]["share_of_probe_video"])   # dominant by time
            roof = dict(roof_all[dom], kernel=dom, regime="one eager video, tile choice of %s" % (
                "one video in flight" if args.probe_one_lane else "the timed region (%d in flight)" % NF))
        roof_timed = None
        tdf = os.path.join(ROOT, TIMED_DURATIONS_FILE)
        if (os.path.exists(tdf) and (args.frames, args.size, N_STEPS, G_STEPS) == (16, 512, 30, 18) and use_graphs
                and not args.sparsectrl and not args.probe_one_lane):
            try:
                roof_timed = timed_roofline(roof_all, json.load(open(tdf)), NF, VB)
            except Exception as e:   # noqa: BLE001  (a stale / malformed profile must not cost the record)
                roof_timed = {"error": "%s: %s" % (type(e).__name__, e)}
        hbm = dict(peak_allocated_gib=timed_allocated, peak_reserved_gib=timed_reserved, videos_in_flight=NF,
                   whole_process_peak_reserved_gib=torch.cuda.max_memory_reserved(dev) / 2 ** 30,
                   note="torch caching allocator during the TIMED region (weights 2.4 GiB fp16 + packed copies, %d lanes: "
                        "static buffers + one hipGraph pool each); whole_process adds the eager probe video run afterwards" % NF)
        res = {
            "metric": "videos/min (%df x %dx%d SD1.5+AnimateDiff-v3 arch, %d-step DDIM, %d guided, MotionClone guidance)"
                      % (args.frames, args.size, args.size, N_STEPS, G_STEPS),
            "value": videos / (elapsed / 60.0), "unit": "videos/min", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"videos_batched_per_lane": VB, "workload": "BASELINE config %s: %d frames, %dx%d, UNet%s only (extraction + %d guided + %d plain "
                                   "DDIM steps), schedule (%d,%d,%g)"
                                   % ("4 (i2v_rgb + SparseCtrl)" if args.sparsectrl else
                                      {(16, 512): "2 (t2v_object-style)", (32, 768): "5 (long clip)",
                                       (16, 256): "1 shape"}.get((args.frames, args.size), "custom"),
                                      args.frames, args.size, args.size, " + SparseCtrl" if args.sparsectrl else "",
                                      G_STEPS, N_STEPS - G_STEPS, N_STEPS, G_STEPS, G_SCALE),
                       "videos_per_gpu": args.steps, "videos_in_flight_per_gpu": NF, "parallelism": "replicas x%d" % world},
            # latency of one step of one video while `videos_in_flight_per_gpu` videos share the GPU, and the throughput view
            # (wall time per step per video = what the videos/min figure is made of)
            "sec_per_guided_step": sum(gsec) / max(1, len(gsec)), "sec_per_plain_step": sum(psec) / max(1, len(psec)),
            "sec_per_denoise_step": (sum(gsec) + sum(psec)) / max(1, len(gsec) + len(psec)),
            "sec_per_denoise_step_throughput": elapsed / (args.steps * N_STEPS),
            "e2e_tflops_per_gpu": tflop_video * args.steps / elapsed,
            "e2e_frac_of_mfma_peak": tflop_video * args.steps / elapsed / PEAK_FP16_MFMA_TFLOPS,
            "roofline": roof,
            "roofline_timed": roof_timed,
            "roofline_note": "per-launch HIP events around EVERY C-ABI launch of ONE eager video run after the timed region "
                             "(events cannot bracket launches inside a graph replay), tile choice of the timed region; each row's "
                             "`bound` is the roof it is closer to (dense fp16 MFMA 2.5 PF vs HBM 8 TB/s, algorithmic flop / bytes); "
                             "`traffic` = PMC HBM bytes per launch on the shapes listed in roofline_traffic_by_shape (profiles/), null "
                             "where not collected; `roofline_timed`: the dominant GEMM family with the per-launch duration of the TIMED "
                             "regime (graphs, videos in flight) from the committed rocprofv3 kernel trace of this command "
                             "(profiles/kernel_durations_timed.json, profiles/r05_kernel_stats.md)",
            "roofline_by_kernel": roof_all,
            "roofline_coverage_of_probe_video": probe.covered(probe_elapsed) if probe_elapsed else None,
            "roofline_traffic_by_shape": traffic_rows,
            "hbm_footprint": hbm,
            # SURVEY.md 8(f) rank 1, measured OUTSIDE the timed region (BASELINE's metric is the UNet loop): the VAE calls
            # around it - decode_latents of the sampled video and the encode of the reference video - on the same kernels
            "vae": vae_info,
            "warmup_videos_run": warm_videos, "warmup_seconds": warm_seconds,
            "graphs": graph_info,
            "eager": eager_info,
            "three_lanes_x_one_video": lanes_only,
        }
        if world == 1 and not args.no_cpu_baseline:
            del eng, smp, smps
            torch.cuda.empty_cache()
            # the two baseline legs run after the measurement; a failure in one of them must not lose the bench line
            try:
                res["reference_gpu_baseline"] = reference_gpu_baseline(dev) if (args.frames, args.size) == (16, 512) else None
            except Exception as e:   # noqa: BLE001
                res["reference_gpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                res["cpu_baseline"] = cpu_baseline()
            except Exception as e:   # noqa: BLE001
                res["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        else:
            res["cpu_baseline"] = None
        if vae_info and "decode_sec_per_video" in vae_info:
            # BASELINE's metric is the UNet loop; a whole video also needs the reference video's VAE encode and the decode of the
            # result (measured above, one video at a time, nothing overlapped): the conservative end-to-end figure
            per_video = elapsed / args.steps + vae_info["decode_sec_per_video"] + vae_info["encode_sec_per_video"]
            res["videos_per_min_incl_vae"] = world * 60.0 / per_video
        detail = write_detail(res) if not args.no_detail else None
        sys.stdout.flush()
        print(compact_line(res, detail), flush=True)
    if dist is not None:
        dist.barrier()      # rank 0 runs its probe / eager videos after the timed region: the ranks leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
