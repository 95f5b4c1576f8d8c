"""The N>1 path on CPU: two gloo ranks shard the example list, receive the packed weights with one broadcast and
max-reduce their timings (replicas only, no data-path collective)."""
import os
import socket

import torch
import torch.multiprocessing as mp

from motionclone_amd import dist as mcd
from motionclone_amd import spec
from oracle import unet3d_ref as U


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w = mcd.init("gloo")
    cfg = dict(U.TINY_CONFIG)
    total = sum(int(torch.Size(s).numel()) for s in spec.param_shapes(cfg).values())
    flat = torch.zeros(total, dtype=torch.float16)
    if r == 0:
        _, flat = spec.synthetic_state_dict(cfg, seed=1234, device="cpu", flat=flat)
    mcd.broadcast_weights(flat, src=0)
    lines = ["ex%d" % i for i in range(5)]
    mine = mcd.shard_examples(lines, r, w)
    t = mcd.max_over_ranks(1.0 + r)
    path = mcd.representation_path("/tmp/mr", "reference_videos/camera_zoom_in.mp4", r, w)
    # a whole checkpoint (nested containers, mixed dtypes) by one object + one byte-buffer broadcast; only rank 0 "reads" it
    ck = None
    if r == 0:
        g = torch.Generator().manual_seed(5)
        ck = {"state_dict": {"a.weight": torch.randn(7, 3, generator=g).half(), "b": torch.arange(5), "n": {"c": torch.randn(2, 2, generator=g)}},
              "meta": ["v3", 1.5, (torch.ones(3, dtype=torch.uint8),)]}
    shared = mcd.SharedCheckpoints()
    got = shared.load("/nonexistent/ckpt.pt", lambda: ck)
    digest = (float(got["state_dict"]["a.weight"].float().sum()), got["state_dict"]["b"].tolist(), got["meta"][:2],
              float(got["state_dict"]["n"]["c"].sum()), got["meta"][2][0].dtype == torch.uint8, shared.reads, shared.received)
    out.put((r, float(flat.double().abs().sum()), [i for i, _ in mine], t, path, digest))
    torch.distributed.destroy_process_group()


def test_two_rank_replicas_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, sum0, ex0, t0, p0, d0), (r1, sum1, ex1, t1, p1, d1) = res
    assert d0[:5] == d1[:5] and d0[5:] == (1, 0) and d1[5:] == (0, 1)   # same checkpoint; read once, received once
    assert sum0 == sum1 and sum0 > 0          # rank 1 received rank 0's weights
    assert ex0 == [0, 2, 4] and ex1 == [1, 3]  # round-robin sharding, every example exactly once
    assert t0 == t1 == 2.0                     # max over ranks
    assert p0 != p1                            # per-rank representation files


def test_failed_checkpoint_read_reaches_the_waiting_lanes():
    """a reader that raises (wrong path) must not leave the other lanes waiting: the failure is published and re-raised"""
    import threading
    import pytest
    shared = mcd.SharedCheckpoints(timeout=20.0)
    got = {}

    def follower():
        try:
            shared.load("missing.ckpt", lambda: 1 / 0, is_leader=False)
        except BaseException as e:   # noqa: BLE001
            got["exc"] = e
    th = threading.Thread(target=follower)
    th.start()

    def reader():
        raise FileNotFoundError("missing.ckpt")
    with pytest.raises(FileNotFoundError):
        shared.load("missing.ckpt", reader, is_leader=True)
    th.join(timeout=30)
    assert not th.is_alive() and isinstance(got.get("exc"), FileNotFoundError)
    with pytest.raises(FileNotFoundError):     # and stays failed for late comers
        shared.load("missing.ckpt", reader, is_leader=False)
    # a follower whose leader never shows up gives up instead of blocking for ever
    with pytest.raises(TimeoutError):
        mcd.SharedCheckpoints(timeout=0.2).load("never.ckpt", reader, is_leader=False)


def _failing_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    mcd.init("gloo")
    shared = mcd.SharedCheckpoints()

    def reader():
        raise FileNotFoundError("no such checkpoint")
    try:
        shared.load("ckpt", reader)
        out.put((rank, "no error"))
    except Exception as e:   # noqa: BLE001
        out.put((rank, "%s: %s" % (type(e).__name__, e)))
    torch.distributed.destroy_process_group()


def test_failed_checkpoint_read_on_rank0_raises_on_every_rank_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0].startswith("FileNotFoundError") and "rank 0 could not read" in res[1] and "FileNotFoundError" in res[1]
