"""Reading checkpoint files (reference util.py:115-215, unet.py:477-515): one place, so that a multi-GPU job reads every
file ONCE.  Stand-alone it is a plain torch.load / safetensors read onto the host.  Under `motionclone_amd.launch` a
`dist.SharedCheckpoints` is installed: rank 0 reads the file, the other ranks receive it by one RCCL / gloo broadcast (the
single collective of the path), and the lanes of a process share the loaded tensors."""
import os

import torch

_shared = None      # dist.SharedCheckpoints while a launcher job runs
_real_torch_load = torch.load


def install(shared):
    global _shared
    _shared = shared


def _copy_containers(obj):
    """fresh dicts / lists around the SAME tensors: callers pop / re-key state dicts, lanes must not see each other's edits"""
    if isinstance(obj, dict):
        return {k: _copy_containers(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_copy_containers(v) for v in obj]
    return obj


def _read_file(path):
    if path.endswith(".safetensors"):
        from safetensors import safe_open
        sd = {}
        with safe_open(path, framework="pt", device="cpu") as f:
            for key in f.keys():
                sd[key] = f.get_tensor(key)
        return sd
    return _real_torch_load(path, map_location="cpu")


def read(path):
    """checkpoint file -> nested containers of CPU tensors"""
    path = os.fspath(path)
    if _shared is None:
        return _read_file(path)
    from . import lanes
    leader = lanes.lane_index() in (None, 0)
    return _copy_containers(_shared.load(os.path.abspath(path), lambda: _read_file(path), is_leader=leader))
