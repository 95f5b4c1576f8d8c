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


# ---- checkpoint files: read once (rank 0), broadcast, shared by the lanes of a process ------------------------------------
def _pack(obj, tensors):
    """nested dict / list / tuple with the tensors replaced by (index, shape, dtype) stubs; tensors collected in order"""
    if isinstance(obj, torch.Tensor):
        tensors.append(obj.detach().cpu().contiguous())
        return ("__tensor__", len(tensors) - 1, tuple(obj.shape), obj.dtype)
    if isinstance(obj, dict):
        return {k: _pack(v, tensors) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_pack(v, tensors) for v in obj)
    return obj


def _unpack(obj, flat, offsets):
    if isinstance(obj, tuple) and len(obj) == 4 and obj[0] == "__tensor__":
        _, i, shape, dtype = obj
        n = int(torch.Size(shape).numel()) * torch.empty((), dtype=dtype).element_size()
        return flat[offsets[i]:offsets[i] + n].view(dtype).view(shape)
    if isinstance(obj, dict):
        return {k: _unpack(v, flat, offsets) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_unpack(v, flat, offsets) for v in obj]
    if isinstance(obj, tuple):
        return tuple(_unpack(v, flat, offsets) for v in obj)
    return obj


def broadcast_state(obj=None, src=0, device=None):
    """One object + ONE byte-buffer broadcast for a whole checkpoint (nested containers of tensors): `obj` is read on `src`
    only, every other rank passes None and gets an equal copy (CPU tensors, views into one buffer).  `device`: where the
    buffer travels (RCCL needs device memory: "cuda"; gloo: None = host)."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return obj
    rank = dist.get_rank()
    meta = [None]
    flat = None
    if rank == src:
        tensors = []
        skel = _pack(obj, tensors)
        offsets, off = [], 0
        for t in tensors:
            offsets.append(off)
            off += (t.numel() * t.element_size() + 15) // 16 * 16
        flat = torch.empty(max(off, 16), dtype=torch.uint8)
        for t, o in zip(tensors, offsets):
            flat[o:o + t.numel() * t.element_size()] = t.view(-1).view(torch.uint8)
        meta = [(skel, offsets, flat.numel())]
    dist.broadcast_object_list(meta, src=src)
    skel, offsets, nbytes = meta[0]
    if rank != src:
        flat = torch.empty(nbytes, dtype=torch.uint8)
    if device is not None:
        buf = flat.to(device)
        dist.broadcast(buf, src=src)
        flat = buf.cpu()
    else:
        dist.broadcast(flat, src=src)
    return obj if rank == src else _unpack(skel, flat, offsets)


class _Failed:
    """cache entry of a checkpoint whose read failed: waiters re-raise instead of waiting for ever"""

    def __init__(self, exc):
        self.exc = exc


class SharedCheckpoints:
    """`load(key, reader)`: the checkpoint file `key` is READ ONCE PER JOB - by rank 0 - and reaches the other ranks by
    `broadcast_state` (the one collective of the path); inside a process it is loaded once and shared by all
    launcher lanes.  Collectives are only ever issued by lane 0's thread (every rank's lane 0 walks the script in the same
    order, so the broadcasts pair up); other lanes wait until lane 0 has published the key.

    Failures are published too: if the reader raises (a wrong checkpoint path), the exception is cached, the waiting
    lanes re-raise it, and - with several ranks - rank 0 sends an ok / error header BEFORE the payload so that the other
    ranks raise instead of blocking in the broadcast.  `timeout` (seconds) bounds a lane's wait as a backstop."""

    def __init__(self, device=None, timeout=1800.0):
        import threading
        self._cv = threading.Condition()
        self._cache = {}
        self._device = device
        self._timeout = timeout
        self.reads = 0          # files actually read from disk by this process
        self.received = 0       # files received from rank 0

    def _result(self, key):
        ent = self._cache[key]
        if isinstance(ent, _Failed):
            raise ent.exc
        return ent

    def load(self, key, reader, is_leader=True):
        with self._cv:
            if key in self._cache:
                return self._result(key)
            if not is_leader:
                if not self._cv.wait_for(lambda: key in self._cache, timeout=self._timeout):
                    raise TimeoutError("checkpoint %r was not published by lane 0 within %.0f s" % (key, self._timeout))
                return self._result(key)
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        try:
            if not multi or dist.get_rank() == 0:
                err = None
                try:
                    obj = reader()
                    self.reads += 1
                except BaseException as e:   # noqa: BLE001 - forwarded to the other ranks / lanes, then re-raised
                    err = e
                if multi:
                    dist.broadcast_object_list([None if err is None else "%s: %s" % (type(err).__name__, err)], src=0)
                if err is not None:
                    raise err
                if multi:
                    broadcast_state(obj, 0, self._device)
            else:
                hdr = [None]
                dist.broadcast_object_list(hdr, src=0)
                if hdr[0] is not None:
                    raise RuntimeError("rank 0 could not read checkpoint %r: %s" % (key, hdr[0]))
                obj = broadcast_state(None, 0, self._device)
                self.received += 1
        except BaseException as e:   # noqa: BLE001
            with self._cv:
                self._cache[key] = _Failed(e)
                self._cv.notify_all()
            raise
        with self._cv:
            self._cache[key] = obj
            self._cv.notify_all()
        return obj
