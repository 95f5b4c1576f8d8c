"""Multi-GPU launcher around the (unmodified) entry scripts: replicas only.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        -m motionclone_amd.launch t2v_video_sample.py --inference_config configs/t2v_camera.yaml \\
        --examples configs/t2v_camera.jsonl [any other argument of the script]

One process per GPU.  Rank r runs the script's own `main(args)` on the lines i of the examples file with
i mod world == r (the reference walks them sequentially, t2v_video_sample.py:75-105), pinned to GPU LOCAL_RANK through
the script's `--visible_gpu`, writing its motion representations under `<dir>/rank<r>/` (several lines share a reference
video and the reference overwrites `<stem>.pt` per example, :89).  Every rank loads the checkpoints itself; the only
collective is a max-reduce of the wall time at the end (RCCL: backend "nccl"; gloo without GPUs).

Serial-RNG fidelity: the scripts seed the GLOBAL generator once (`set_all_seed(42)`) and the VAE
posterior of every example draws from it, so an example's motion representation depends on how many examples ran
before it.  With `--serial-rng` (default on) a rank burns, for every line it skips, the draws that line would have made
(one `[L, 4, H/8, W/8]` normal tensor for the reference video, plus `[n_images, 4, H/8, W/8]` for i2v condition images),
so that the sharded run reproduces the single-process run bit for bit."""
import builtins
import io
import json
import os
import runpy
import sys
import time

import torch

from . import dist as mcd


def _arg(argv, names, default=None):
    for i, a in enumerate(argv):
        for n in names:
            if a == n and i + 1 < len(argv):
                return argv[i + 1]
            if a.startswith(n + "="):
                return a.split("=", 1)[1]
    return default


def _set_arg(argv, name, value):
    out, skip = [], False
    for i, a in enumerate(argv):
        if skip:
            skip = False
            continue
        if a == name:
            skip = True
            continue
        if a.startswith(name + "="):
            continue
        out.append(a)
    return out + [name, str(value)]


def _condition_images_are_encoded(inference_config):
    """True iff the SparseCtrl variant named by the inference config's `controlnet_config` uses the simplified (latent)
    condition embedding.  Read with yaml (OmegaConf files are plain yaml); a config that cannot be read counts as the
    latent variant, the reference's i2v_rgb default."""
    try:
        import yaml
        with open(inference_config) as f:
            cfg = yaml.safe_load(f) or {}
        cpath = cfg.get("controlnet_config")
        if not cpath:
            return True
        if not os.path.isabs(cpath) and not os.path.exists(cpath):
            cpath = os.path.join(os.path.dirname(os.path.abspath(inference_config)), os.pardir, cpath)
        with open(cpath) as f:
            ccfg = yaml.safe_load(f) or {}
        kw = ccfg.get("controlnet_additional_kwargs", ccfg)
        return bool(kw.get("use_simplified_condition_embedding", True))
    except (OSError, ValueError, AttributeError, ImportError):
        return True


class _ShardedLines(io.StringIO):
    """file object over the examples file that yields only this rank's lines; before each of them it burns the global-RNG
    draws of the lines skipped since the previous one (see module docstring)"""

    def __init__(self, lines, rank, world, burn):
        super().__init__("")
        self._items = [(i, ln) for i, ln in enumerate(lines) if ln.strip()]
        self._rank, self._world, self._burn = rank, world, burn

    def __iter__(self):
        for i, ln in self._items:
            if i % self._world == self._rank:
                yield ln
            elif self._burn is not None:
                self._burn(json.loads(ln))


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit(__doc__)
    serial_rng = True
    if "--no-serial-rng" in argv:
        argv.remove("--no-serial-rng")
        serial_rng = False
    vae_scale = int(_arg(argv, ["--vae-scale"], 8))
    if "--vae-scale" in argv:
        i = argv.index("--vae-scale")
        del argv[i:i + 2]
    script, sargv = argv[0], argv[1:]
    # Pin the GPU BEFORE anything initialises the HIP runtime: the runtime reads CUDA_VISIBLE_DEVICES once, and both
    # torch.cuda.is_available() and the scripts' own late `os.environ["CUDA_VISIBLE_DEVICES"] = args.visible_gpu` (inside
    # main()) come after it otherwise - every rank would then sit on device 0 and RCCL would refuse the duplicate devices.
    local = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    multi = int(os.environ.get("WORLD_SIZE", "1")) > 1
    if multi and os.environ.get("MC_LAUNCH_NO_PIN", "0") != "1":
        visible = os.environ.get("CUDA_VISIBLE_DEVICES")
        ids = [v for v in visible.split(",") if v] if visible else None
        os.environ["CUDA_VISIBLE_DEVICES"] = ids[local % len(ids)] if ids else str(local)
    rank, world = mcd.init()
    is_i2v = "i2v" in os.path.basename(script)
    examples = _arg(sargv, ["--examples"], "configs/i2v_sketch.jsonl" if is_i2v else "configs/t2v_camera.jsonl")
    rep_dir = _arg(sargv, ["--motion-representation-save-dir"], "motion_representation/")
    L = int(_arg(sargv, ["--L"], 16))
    H, W = int(_arg(sargv, ["--H"], 512)), int(_arg(sargv, ["--W"], 512))
    with open(examples) as f:
        lines = f.readlines()
    mine = mcd.shard_examples([ln for ln in lines if ln.strip()], rank, world)
    if world > 1:
        sargv = _set_arg(sargv, "--motion-representation-save-dir", os.path.join(rep_dir, "rank%d" % rank))
        if torch.cuda.is_available():
            # the script assigns CUDA_VISIBLE_DEVICES = args.visible_gpu inside main(): hand it the value set above, so
            # the assignment is a no-op whether or not the runtime is already up
            sargv = _set_arg(sargv, "--visible_gpu", os.environ["CUDA_VISIBLE_DEVICES"])
    # Only the latent-condition SparseCtrl (use_simplified_condition_embedding: true, sparsectrl/latent_condition.yaml)
    # VAE-encodes the condition images and so draws from the global generator (motionclone_functions.py:122-126); the
    # pixel-condition variant (image_condition.yaml, :127-128) draws nothing for them.
    encodes_condition = False
    if is_i2v:
        encodes_condition = _condition_images_are_encoded(_arg(sargv, ["--inference_config"], "configs/i2v_sketch.yaml"))

    def burn(example):
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        torch.randn((L, 4, H // vae_scale, W // vae_scale), device=dev, dtype=torch.float16)
        n_img = len(example.get("condition_image_paths", ())) if encodes_condition else 0
        if n_img:
            torch.randn((n_img, 4, H // vae_scale, W // vae_scale), device=dev, dtype=torch.float16)

    real_open = builtins.open
    ex_abs = os.path.abspath(examples)

    def sharded_open(path, *a, **k):
        if world > 1 and isinstance(path, (str, os.PathLike)) and os.path.abspath(path) == ex_abs and (not a or "r" in a[0]):
            return _ShardedLines(lines, rank, world, burn if serial_rng else None)
        return real_open(path, *a, **k)

    builtins.open = sharded_open
    t0 = time.perf_counter()
    try:
        sys.argv = [script] + sargv
        runpy.run_path(script, run_name="__main__")
    finally:
        builtins.open = real_open
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    slowest = mcd.max_over_ranks(dt, device="cuda" if torch.cuda.is_available() and world > 1 else "cpu")
    if rank == 0:
        n = len([ln for ln in lines if ln.strip()])
        print(json.dumps(dict(examples=n, world=world, seconds=slowest, videos_per_min=60.0 * n / slowest,
                              examples_of_rank0=[i for i, _ in mine])))
    if world > 1 and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
