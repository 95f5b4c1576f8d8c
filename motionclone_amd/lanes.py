"""Several examples in flight inside ONE process / GPU ("lanes"): what `motionclone_amd.launch --lanes K` runs on.

A lane is a host thread with its own HIP stream that executes the UNMODIFIED entry script on its share of the examples file.
Independent (prompt, reference-video) examples are the unit of parallelism of this workload; K of them on one
GPU fill each other's kernel tails and under-filled launches (bench.py: +14-20 % videos/min for three in flight).

The one piece of shared state the scripts rely on is torch's GLOBAL generator: `set_all_seed(42)` once, then every example's
VAE posterior draw comes from it, so an example's result depends on how many examples ran before it.
Threads would interleave those draws, so inside a lane the "global" stream is a LANE-PRIVATE generator: `set_all_seed` seeds
it, the drop-in VAE draws from it, and the launcher burns the draws of the examples the lane skips - each lane reproduces the
serial run's stream position for every example it owns, and the results stay bit-identical to the single-process run."""
import threading

import torch

_LOCAL = threading.local()


def begin(lane, n_lanes, device):
    """mark the calling thread as lane `lane` of `n_lanes`; its serial-RNG stream lives on `device`"""
    _LOCAL.lane, _LOCAL.n, _LOCAL.dev, _LOCAL.gen, _LOCAL.warm = lane, n_lanes, torch.device(device), None, False


def warmed_up():
    """called by the launcher when the lane's first example is done: from here on the lanes run concurrently"""
    _LOCAL.warm = True


def may_capture():
    """hipGraph capture is only done while a lane runs ALONE (its first example: the launcher serialises those).  ROCm 7.2
    rejects synchronising calls of ANY host thread while a capture is open (hipErrorStreamCaptureUnsupported, also in
    thread-local capture mode), so a step whose graph is missing once the lanes run concurrently is issued eagerly."""
    return not active() or not getattr(_LOCAL, "warm", False)


def end():
    _LOCAL.lane = None


def active():
    return getattr(_LOCAL, "lane", None) is not None


def lane_index():
    return getattr(_LOCAL, "lane", None)


def seed(value):
    """set_all_seed inside a lane: (re)seed the lane's private stream the way torch.manual_seed seeds the global one"""
    if active():
        g = torch.Generator(device=_LOCAL.dev)
        g.manual_seed(int(value))
        _LOCAL.gen = g


def serial_generator():
    """None outside a lane (= torch's global generator, the reference's behaviour); the lane's stream inside one"""
    if not active():
        return None
    if _LOCAL.gen is None:   # a script that never seeds: same default the global generator would have had is unknowable
        seed(torch.initial_seed())
    return _LOCAL.gen
