"""Multi-GPU launcher around the (unmodified) entry scripts: replicas only.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        -m motionclone_amd.launch t2v_video_sample.py --inference_config configs/t2v_camera.yaml \\
        --examples configs/t2v_camera.jsonl [any other argument of the script]

One process per GPU.  Rank r runs the script's own `main(args)` on the lines i of the examples file with
i mod world == r (the reference walks them sequentially, t2v_video_sample.py:75-105), pinned to GPU LOCAL_RANK through
the script's `--visible_gpu`, writing its motion representations under `<dir>/rank<r>/` (several lines share a reference
video and the reference overwrites `<stem>.pt` per example, :89).  Checkpoint files are read from disk by rank 0 only and reach
the other ranks by one broadcast per file (RCCL: backend "nccl"; gloo without GPUs; `--no-broadcast-weights`: every rank reads
them itself); the only other collective is a max-reduce of the wall time at the end.

`--lanes K` (default 1): K examples IN FLIGHT per process, each in its own host thread + HIP stream running the same unmodified
script on every K-th of the rank's lines (motionclone_amd/lanes.py): the kernels of one example leave CUs idle in tails and in
the small 16x16 / 8x8-level launches, which the other lanes fill - the regime bench.py times (`--inflight`).  The DDIM steps
replay from hipGraphs (captured per lane at first use).  Every lane loads its own copy of the checkpoints (HBM is 288 GB).

Serial-RNG fidelity: the scripts seed the GLOBAL generator once (`set_all_seed(42)`) and the VAE
posterior of every example draws from it, so an example's motion representation depends on how many examples ran
before it.  With `--serial-rng` (default on) a rank burns, for every line it skips, the draws that line would have made
(one `[L, 4, H/8, W/8]` normal tensor for the reference video, plus `[n_images, 4, H/8, W/8]` for i2v condition images),
so that the sharded run reproduces the single-process run bit for bit.  Lanes shard the same way (line i -> rank i mod world,
lane (i div world) mod K) and each lane keeps a PRIVATE copy of the "global" stream (lanes.py), so `--lanes K` is bit-identical
to the serial run as well (tests/test_entry_scripts.py)."""
import argparse
import builtins
import io
import json
import os
import runpy
import sys
import threading
import time

import torch

from . import dist as mcd
from . import lanes as mcl


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

    def __init__(self, lines, rank, world, burn, first_begin=None, first_end=None):
        super().__init__("")
        self._items = [(i, ln) for i, ln in enumerate(lines) if ln.strip()]
        self._rank, self._world, self._burn = rank, world, burn
        self._first_begin, self._first_end = first_begin, first_end     # lanes: the first example of a lane runs alone

    def __iter__(self):
        n = 0
        try:
            for i, ln in self._items:
                if i % self._world == self._rank:
                    if n == 0 and self._first_begin:
                        self._first_begin()
                    if n == 1 and self._first_end:       # asked for the second line = the first example is finished
                        self._first_end()
                    n += 1
                    yield ln
                elif self._burn is not None:
                    self._burn(json.loads(ln))
        finally:
            if n == 0 and self._first_begin:             # a lane without examples still takes its turn
                self._first_begin()
            if n <= 1 and self._first_end:
                self._first_end()


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
    broadcast_weights = True
    if "--no-broadcast-weights" in argv:
        argv.remove("--no-broadcast-weights")
        broadcast_weights = False
    n_lanes = int(_arg(argv, ["--lanes"], 1))
    if "--lanes" in argv:
        i = argv.index("--lanes")
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
    on_gpu = torch.cuda.is_available()
    if world > 1 and on_gpu:
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
        dev = "cuda" if on_gpu else "cpu"
        gen = mcl.serial_generator()      # the lane's private stream, or None = torch's global generator
        torch.randn((L, 4, H // vae_scale, W // vae_scale), device=dev, dtype=torch.float16, generator=gen)
        n_img = len(example.get("condition_image_paths", ())) if encodes_condition else 0
        if n_img:
            torch.randn((n_img, 4, H // vae_scale, W // vae_scale), device=dev, dtype=torch.float16, generator=gen)

    # virtual rank of (rank, lane): line i belongs to it iff i mod (world K) == rank + world lane
    vworld = world * n_lanes
    sharded = vworld > 1
    real_open = builtins.open
    ex_abs = os.path.abspath(examples)

    def sharded_open(path, *a, **k):
        if sharded and isinstance(path, (str, os.PathLike)) and os.path.abspath(path) == ex_abs and (not a or "r" in a[0]):
            vrank = rank + world * (mcl.lane_index() or 0)
            fb, fe = (warm_begin, warm_end) if n_lanes > 1 else (None, None)
            if n_lanes > 1:      # the lane has built its models (their H2D copies synchronise): it is ready for the warm-up turns
                with warm_turn:
                    warm_state["ready"] += 1
                    warm_turn.notify_all()
            return _ShardedLines(lines, vrank, vworld, burn if serial_rng else None, fb, fe)
        return real_open(path, *a, **k)

    # A lane's FIRST example runs alone: it captures the lane's step graphs, and ROCm 7.2 rejects synchronising calls of any
    # other host thread while a capture is open.  Turns are taken in lane order; when every lane has had its turn they all
    # continue concurrently (replays only; lanes.may_capture).
    # No lane takes its turn before EVERY lane has opened the examples file, i.e. has finished load_state_dict / .to(cuda):
    # those are synchronising calls too and would hit lane 0's open capture otherwise.  A lane that dies releases the others.
    warm_turn = threading.Condition()
    warm_state = dict(turn=0, ready=0)

    def warm_begin():
        with warm_turn:
            warm_turn.wait_for(lambda: errors or (warm_state["ready"] >= n_lanes
                                                  and warm_state["turn"] == (mcl.lane_index() or 0)))
            if errors:
                raise RuntimeError("lane %d failed: %r" % errors[0])

    def warm_end():
        if on_gpu:
            torch.cuda.current_stream().synchronize()
        mcl.warmed_up()
        with warm_turn:
            warm_state["turn"] += 1
            warm_turn.notify_all()
            warm_turn.wait_for(lambda: errors or warm_state["turn"] >= n_lanes)

    def lane_argv(lane):
        out = sargv
        if sharded:
            sub = "rank%d" % rank if n_lanes == 1 else "rank%d_lane%d" % (rank, lane)
            out = _set_arg(out, "--motion-representation-save-dir", os.path.join(rep_dir, sub))
        return out

    # Checkpoint files are read once per JOB: rank 0 reads, the other ranks receive them by one broadcast each, the lanes of a
    # process share them (checkpoints.py / dist.SharedCheckpoints).  The scripts' own direct torch.load calls (the SparseCtrl
    # checkpoint, i2v_video_sample.py) go the same way; the per-example motion-representation files do not.
    from . import checkpoints as mck
    share_ckpt = broadcast_weights and (world > 1 or n_lanes > 1)
    real_torch_load = torch.load
    rep_abs = os.path.abspath(rep_dir)

    def shared_torch_load(f, *a, **k):
        if share_ckpt and isinstance(f, (str, os.PathLike)) and not os.path.abspath(os.fspath(f)).startswith(rep_abs):
            return mck.read(f)
        return real_torch_load(f, *a, **k)

    # the scripts read sys.argv through argparse; lanes need different argument lists, so parse_known_args looks at a
    # thread-local list first
    targv = threading.local()
    real_parse = argparse.ArgumentParser.parse_known_args

    def parse_known_args(self, args=None, namespace=None):
        if args is None and getattr(targv, "argv", None) is not None:
            args = list(targv.argv)
        return real_parse(self, args, namespace)

    errors = []

    def run_lane(lane):
        try:
            if n_lanes > 1:
                mcl.begin(lane, n_lanes, "cuda" if on_gpu else "cpu")
                if on_gpu:
                    torch.cuda.set_stream(torch.cuda.Stream())
            targv.argv = lane_argv(lane)
            runpy.run_path(script, run_name="__main__")
            if on_gpu:
                torch.cuda.current_stream().synchronize()
        except BaseException as e:   # noqa: BLE001 - reported by the main thread
            errors.append((lane, e))
            with warm_turn:          # lanes waiting for their warm-up turn must not wait for a dead one
                warm_turn.notify_all()
        finally:
            mcl.end()

    builtins.open = sharded_open
    argparse.ArgumentParser.parse_known_args = parse_known_args
    shared = None
    if share_ckpt:
        shared = mcd.SharedCheckpoints(device="cuda" if on_gpu and world > 1 else None)
        mck.install(shared)
        torch.load = shared_torch_load
    t0 = time.perf_counter()
    try:
        sys.argv = [script] + lane_argv(0)
        if n_lanes == 1:
            run_lane(0)
        else:
            from . import ops
            ops.set_gemm_share(n_lanes)     # tile / split-K choice for n_lanes launch sequences in flight (before any graph capture)
            threads = [threading.Thread(target=run_lane, args=(k,), name="lane%d" % k) for k in range(n_lanes)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        if errors:
            raise errors[0][1]
    finally:
        builtins.open = real_open
        argparse.ArgumentParser.parse_known_args = real_parse
        torch.load = real_torch_load
        mck.install(None)
    if on_gpu:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    slowest = mcd.max_over_ranks(dt, device="cuda" if on_gpu and world > 1 else "cpu")
    if rank == 0:
        n = len([ln for ln in lines if ln.strip()])
        print(json.dumps(dict(examples=n, world=world, lanes=n_lanes, seconds=slowest, videos_per_min=60.0 * n / slowest,
                              examples_of_rank0=[i for i, _ in mine],
                              checkpoint_files_read_by_rank0=shared.reads if shared else None)))
    elif shared is not None:
        print(json.dumps(dict(rank=rank, checkpoint_files_read_from_disk=shared.reads, received_by_broadcast=shared.received)))
    if world > 1 and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
