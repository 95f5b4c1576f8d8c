"""Multi-GPU layer: replicas only.  Independent (prompt, reference-video) examples are sharded
round-robin over one process per GPU; the single data collective is a broadcast of the packed fp16 weight buffer
from rank 0 (RCCL over xGMI on the GPU box - backend "nccl" - or gloo in CPU tests); timing is max-reduced."""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise from the torchrun environment; returns (rank, world).  No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def shard_examples(lines, rank, world):
    """line i -> rank i mod world (the reference processes them sequentially: t2v_video_sample.py:75-105)"""
    return [(i, ln) for i, ln in enumerate(lines) if i % world == rank]


def representation_path(save_dir, video_path, rank, world):
    """several lines share a reference video and the reference overwrites motion_representation/<stem>.pt per
    example (t2v_video_sample.py:89): give every rank its own file so replicas never race."""
    stem = os.path.splitext(os.path.basename(video_path))[0]
    name = stem + (".pt" if world == 1 else ".rank%d.pt" % rank)
    return os.path.join(save_dir, name)


def broadcast_weights(flat, src=0):
    """one collective for all parameters: they are views into a single flat buffer (spec.synthetic_state_dict)"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src=src)
    return flat


def max_over_ranks(value, device="cpu"):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([float(value)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return float(value)
