"""Worker of tests/test_multigpu.py: ONE rank of an N-rank replica job, started by torch.distributed.run.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         tests/multirank_worker.py <out_dir> [--cpu]

Every rank pins its GPU from LOCAL_RANK, receives the packed weights by ONE broadcast into DEVICE memory (RCCL on GPUs) and
a checkpoint file through `dist.SharedCheckpoints` (read by rank 0 only; on GPUs the byte buffer travels through device
memory), runs the guided DDIM loop on ITS examples (example i -> rank i mod N) and writes their latents.  Rank 0 then
repeats ALL examples serially in the same process; the test compares the sharded results with that serial run bit for bit.
`--cpu`: gloo + the host simulator of the kernels (what this container can run); otherwise nccl (= RCCL) + the HIP library."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_EXAMPLES = 2
SEEDS = [42, 2026]


def run_example(eng, dev, i, cfg):
    from motionclone_amd.sampler import MotionCloneSampler
    F, H = 4, 8
    g = torch.Generator().manual_seed(SEEDS[i])
    lat = torch.randn(1, 4, F, H, H, generator=g).half().to(dev)
    text = torch.randn(2, 7, cfg["cross_attention_dim"], generator=g).half().to(dev)
    vid = (0.18215 * torch.randn(1, 4, F, H, H, generator=g)).half().to(dev)
    noise = torch.randn(1, 4, F, H, H, generator=g).half().to(dev)
    smp = MotionCloneSampler(eng, num_inference_steps=2, guidance_steps=1, guidance_scale=0.3, cfg_scale=7.5,
                             motion_guidance_weight=2000.0, warm_up_steps=10, cool_up_steps=10)
    rep = smp.extract(vid, noise, text[0:1], add_noise_step=400)
    return smp.sample(lat, text, rep).float().cpu()


def main():
    out_dir = sys.argv[1]
    cpu = "--cpu" in sys.argv
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    from motionclone_amd import dist as mcd
    from motionclone_amd import lib, spec
    from motionclone_amd.engine import UNet3DEngine
    from oracle import unet3d_ref as U     # TINY_CONFIG only (test infrastructure; this file lives under tests/)
    if cpu:
        from motionclone_amd import build
        lib.use_library_for_tests(build.build_emu())
        dev = torch.device("cpu")
    else:
        assert torch.cuda.device_count() >= world, (torch.cuda.device_count(), world)
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        lib.load()
    mcd.init("gloo" if cpu else "nccl")
    cfg = dict(U.TINY_CONFIG)
    # ---- weights: one broadcast of the flat buffer, in device memory --------------------------------------------------
    total = sum(int(torch.Size(s).numel()) for s in spec.param_shapes(cfg).values())
    flat = torch.zeros(total, dtype=torch.float16, device=dev)
    if rank == 0:
        _, flat = spec.synthetic_state_dict(cfg, seed=1234, device=dev, flat=flat)
    mcd.broadcast_weights(flat, src=0)
    sd, off = {}, 0
    for name, shape in spec.param_shapes(cfg).items():
        n = int(torch.Size(shape).numel())
        sd[name] = flat[off:off + n].view(shape)
        off += n
    # ---- a checkpoint file: read by rank 0 only, broadcast (device memory on GPUs) ------------------------------------
    ck_path = os.path.join(out_dir, "extra.ckpt")
    if rank == 0:
        torch.save({"state_dict": {"w": torch.arange(12, dtype=torch.float32).reshape(3, 4), "h": torch.ones(5).half()},
                    "tag": "v3"}, ck_path)
    shared = mcd.SharedCheckpoints(device=None if cpu else "cuda")
    reads = {"n": 0}

    def reader():
        reads["n"] += 1
        return torch.load(ck_path)
    ck = shared.load(ck_path, reader)
    assert ck["tag"] == "v3" and float(ck["state_dict"]["w"].sum()) == 66.0 and ck["state_dict"]["h"].dtype == torch.float16
    # ---- this rank's examples -------------------------------------------------------------------------------------------
    eng = UNet3DEngine(sd, cfg, dev)
    mine = [i for i, _ in mcd.shard_examples(list(range(N_EXAMPLES)), rank, world)]
    for i in mine:
        torch.save(run_example(eng, dev, i, cfg), os.path.join(out_dir, "sharded_%d.pt" % i))
    t = mcd.max_over_ranks(1.0 + rank, device=dev)
    info = dict(rank=rank, world=world, local_rank=local, examples=mine, checkpoint_reads=reads["n"],
                received=shared.received, max_over_ranks=t, weights_abs_sum=float(flat.double().abs().sum()),
                device=str(dev), current_device=None if cpu else torch.cuda.current_device(),
                device_name=None if cpu else torch.cuda.get_device_name(local),
                backend=torch.distributed.get_backend() if world > 1 else None)
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(info, f)
    if world > 1:
        torch.distributed.barrier()
    if rank == 0:     # the serial run of every example, same process, same engine
        for i in range(N_EXAMPLES):
            torch.save(run_example(eng, dev, i, cfg), os.path.join(out_dir, "serial_%d.pt" % i))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    print("WORKER_OK rank %d" % rank)


if __name__ == "__main__":
    main()
