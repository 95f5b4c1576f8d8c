"""The N > 1 path end to end (replicas only - example i -> rank i mod N, one weight broadcast, checkpoint files
read once, no per-step collective), launched exactly as the driver launches bench.py: `python -m torch.distributed.run`.

  * here (no GPU): two gloo ranks on the host simulator - the SAME worker code path (`tests/multirank_worker.py --cpu`);
  * on a box with >= 2 MI355X (`-m gpu`, skipped on a 1-GPU box): two RCCL ranks - per-rank device ids, weights and checkpoint
    broadcast through DEVICE memory, every sharded video bit-identical to the serial run - and `python bench.py --gpus 2`
    WITHOUT a launcher (it spawns its own ranks), so that the driver's scaling leg cannot fail on plumbing."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "multirank_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world, out_dir, extra):
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), WORKER, str(out_dir)] + extra
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=1500)
    assert r.returncode == 0 and r.stdout.count("WORKER_OK") == world, r.stdout[-4000:]


def _check(out_dir, world, gpu):
    infos = [json.load(open(os.path.join(str(out_dir), "rank%d.json" % r))) for r in range(world)]
    assert [i["examples"] for i in infos] == [[e for e in range(2) if e % world == r] for r in range(world)]
    assert infos[0]["checkpoint_reads"] == 1 and all(i["checkpoint_reads"] == 0 and i["received"] == 1 for i in infos[1:])
    assert len({i["weights_abs_sum"] for i in infos}) == 1 and infos[0]["weights_abs_sum"] > 0
    assert all(i["max_over_ranks"] == float(world) for i in infos)
    if gpu:
        assert [i["current_device"] for i in infos] == list(range(world)), "a rank is not on its own GPU"
        assert all(i["backend"] == "nccl" and i["device"] == "cuda:%d" % i["local_rank"] for i in infos)
    for e in range(2):
        a = torch.load(os.path.join(str(out_dir), "sharded_%d.pt" % e))
        b = torch.load(os.path.join(str(out_dir), "serial_%d.pt" % e))
        assert torch.isfinite(a).all() and torch.equal(a, b), "example %d differs from the serial run" % e


def test_two_gloo_ranks_on_the_simulator_reproduce_the_serial_run(tmp_path):
    _launch(2, tmp_path, ["--cpu"])
    _check(tmp_path, 2, gpu=False)


@pytest.mark.gpu
def test_two_rccl_ranks_reproduce_the_serial_run(tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (this box has %d)" % torch.cuda.device_count())
    _launch(2, tmp_path, [])
    _check(tmp_path, 2, gpu=True)


@pytest.mark.gpu
def test_bench_gpus_2_without_a_launcher(tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (this box has %d)" % torch.cuda.device_count())
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size",
                        "256", "--ddim-steps", "6", "--guided-steps", "3", "--inflight", "2", "--no-cpu-baseline", "--no-vae",
                        "--no-detail"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=1500)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-4000:]
    got = json.loads(lines[0])
    assert got["n_gpus"] == 2 and got["value"] > 0 and got["scaling"] == "weak" and got["config"]["parallelism"] == "replicas x2"
