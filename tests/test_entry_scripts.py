"""The drop-in boundary for real: the UNMODIFIED reference entry scripts
/root/reference/t2v_video_sample.py and i2v_video_sample.py run `main(args)` against this repo's `motionclone/`
package (tests/entry_harness.py, a child process: tiny checkpoints / configs / 4-frame video written to a tmp dir,
kernels on the host simulator), and what they produced is checked against the oracle:

  * the motion representation `.pt` written by obtain_motion_representation and read back by sample_video
    (torch.save -> torch.load round trip, motionclone_functions.py:81,154): keys, dtypes, shapes, values / indices;
  * the latents after the step loop vs the oracle loop from the same start (with the SparseCtrl residuals for i2v);
  * the frames handed to imageio vs the oracle VAE decode of those latents;
  * the text embeddings vs the real transformers.CLIPTextModel on the script's own token ids.

Runs only where the reference tree exists (this container)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import guidance_ref as G
from oracle import reference_shim as shim
from oracle import unet3d_ref as U
from oracle import vae_ref as V

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not shim.available(), reason="reference tree not present")


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


N_EXAMPLES = {"t2v": 2, "i2v": 1, "i2v_sketch": 2}
LAST = {"t2v": ("a dog walks 1", 2027), "i2v": ("a cat runs", 42), "i2v_sketch": ("a dog walks 1", 2027)}


def start_harness(kind, work):
    env = dict(os.environ, PYTHONPATH=ROOT)
    return subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "entry_harness.py"), kind, str(work), "--examples",
                             str(N_EXAMPLES[kind])], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env,
                            cwd=str(work))


@pytest.fixture(scope="module")
def runs(tmp_path_factory):
    """every kind's child process is started on first use, so the three script runs overlap (each takes about a minute on
    the host simulator)"""
    procs, cache = {}, {}

    def get(kind):
        if not procs:
            for k in N_EXAMPLES:
                work = tmp_path_factory.mktemp(k)
                procs[k] = (work, start_harness(k, work))
        if kind not in cache:
            work, p = procs[kind]
            out = p.communicate(timeout=1500)[0]
            assert p.returncode == 0 and "ENTRY_OK" in out, out[-4000:]
            cache[kind] = (work, torch.load(os.path.join(str(work), "record.pt")))
        return cache[kind]
    yield get
    for k, (_, p) in procs.items():
        if p.poll() is None:
            p.kill()


@pytest.mark.parametrize("kind", ["t2v", "i2v", "i2v_sketch"])
def test_unmodified_entry_script_matches_oracle(kind, runs):
    import entry_harness as EH
    tmp_path, rec = runs(kind)
    cfg = dict(U.TINY_CONFIG)
    sd = {k: v.half().float() for k, v in U.random_state_dict(cfg, seed=1234).items()}
    N, Gs, gs = EH.STEPS, EH.GUIDED, EH.GSCALE

    # quirk 2: the yaml key is `postive_prompt`, the script reads `positive_prompt` -> the suffix is never appended
    want_names = ["clip_a_cat_runs42_42.mp4"] + ["clip_a_dog_walks_%d%d_%d.mp4" % (n, 2026 + n, 2026 + n)
                                                 for n in range(1, N_EXAMPLES[kind])]
    assert [os.path.basename(p) for p in rec["videos"]] == want_names

    # ---- the .pt written by the script and loaded back by sample_video ------------------------------------------------
    pt = torch.load(os.path.join(str(tmp_path), "motion_representation", "clip.pt"))
    names = ["up_blocks.1.motion_modules.%d.temporal_transformer.transformer_blocks.0.attention_blocks.%d" % (j, a)
             for j in range(3) for a in range(2)]
    assert list(pt) == names
    ex = rec["extract"]
    assert ex["t"] == 400 and ex["noisy"].shape == (1, 4, EH.F, 8, 8)
    res = None
    csd = None
    i2v = kind != "t2v"
    if i2v:
        simplified = kind == "i2v"
        csd = {k: v.half().float() for k, v in U.random_controlnet_state_dict(
            cfg, conditioning_channels=4 if simplified else 3, simplified=simplified).items()}
        c0 = rec["controlnet_calls"][0]
        # latent_condition.yaml: the condition is the VAE latent of the frame; image_condition.yaml: the frame itself in [0, 1]
        assert tuple(c0["cond"].shape) == ((1, 4, EH.F, 8, 8) if simplified else (1, 3, EH.F, EH.px_of(kind), EH.px_of(kind)))
        if not simplified:
            assert 0.0 <= float(c0["cond"].min()) and float(c0["cond"].max()) <= 1.0 and float(c0["cond"][:, :, 0].max()) > 0.5
        assert c0["t"] == 400 and c0["B"] == 1 and abs(c0["scale"] - 0.8) < 1e-6
        assert float(c0["mask"][:, :, 0].min()) == 1.0 and float(c0["mask"][:, :, 1:].abs().max()) == 0.0
        with torch.no_grad():
            res = U.controlnet_forward(csd, cfg, ex["noisy"].shape, 400, ex["text"].float(), c0["cond"].float(),
                                       c0["mask"].float(), 0.8)
    rc = {}
    with torch.no_grad():
        U.unet_forward(sd, cfg, ex["noisy"].float(), 400, ex["text"].float(), only_motion_feature=True, record=rc,
                       down_residuals=res[0] if res else None, mid_residual=res[1] if res else None)
        prob = G.temp_attn_prob(rc, cfg["motion_heads"])
    ref_rep = G.motion_representation(prob)
    for k in names:
        v, i = pt[k]
        assert v.dtype == torch.float16 and i.dtype == torch.uint8 and v.device.type == "cpu"
        assert v.shape == ref_rep[k][0].shape == (4, cfg["motion_heads"], EH.F, 1) and i.shape == v.shape
        assert (v.float() - ref_rep[k][0]).abs().max() < 5e-3
        mism = i != ref_rep[k][1]
        if mism.any():
            alt = torch.gather(prob[k], -1, i.long())
            assert ((ref_rep[k][0] - alt)[mism] < 4e-3).all()

    # ---- the step loop ---------------------------------------------------------------------------------------------------
    lp = rec["loop"]
    assert lp["steps_run"] == N and lp["G"] == Gs and lp["timesteps"] == G.uneven_timesteps(N, Gs, gs).tolist()
    ts = G.uneven_timesteps(N, Gs, gs)
    hp = dict(cfg_scale=7.5, motion_guidance_weight=2000.0, warm_up_steps=10, cool_up_steps=10, guidance_steps=Gs)
    rep_cpu = {k: [a.float(), b] for k, (a, b) in pt.items()}
    x, text = lp["lat0"].float(), lp["text"].float()
    assert i2v == (lp["ctrl"] is not None)
    for s in range(N):
        d = m = None
        if i2v:
            with torch.no_grad():
                d, m = U.controlnet_forward(csd, cfg, (2, 4, EH.F, 8, 8), int(ts[s]), text, lp["ctrl"]["cond"].float(),
                                            lp["ctrl"]["mask"].float(), 0.8)
        if s < Gs:
            x, _ = G.guided_step(sd, cfg, x, s, ts, text, rep_cpu, hp,
                                 res_u=([t[[0]] for t in d], m[[0]]) if d else None,
                                 res_c=([t[[1]] for t in d], m[[1]]) if d else None)
        else:
            x, _ = G.plain_step_full(sd, cfg, x, s, ts, text, 7.5, res=(d, m) if d else None)
    assert rel(lp["last"], x) < 3e-2, rel(lp["last"], x)

    # ---- decoded video handed to imageio.mimwrite --------------------------------------------------------------------
    frames = np.load(rec["videos"][-1] + ".npy")    # the records are those of the last example
    assert frames.dtype == np.uint8 and frames.shape == (EH.F, EH.px_of(kind), EH.px_of(kind), 3)
    vcfg = EH.vae_config_of(kind)
    vsd = {k: v.half().float() for k, v in V.random_state_dict(vcfg, seed=77).items()}
    with torch.no_grad():
        want = V.decode_latents(vsd, vcfg, lp["last"].float())[0].permute(1, 2, 3, 0).numpy()   # f h w c in [0, 1]
    assert np.abs(frames.astype(np.float32) / 255.0 - want).max() < 0.03

    # ---- prompt encoding: [negative_prompt, new_prompt] through the script's tokenizer + the drop-in text encoder -----
    import transformers
    tok = transformers.CLIPTokenizer.from_pretrained(os.path.join(str(tmp_path), "sd"), subfolder="tokenizer")
    ids = tok(["bad quality", LAST[kind][0]], padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
    cc = EH.tiny_clip_config()
    hc = transformers.CLIPTextConfig(**{k: cc[k] for k in cc}, attn_implementation="eager")
    hf = transformers.CLIPTextModel(hc).eval()
    wsd = torch.load(os.path.join(str(tmp_path), "sd", "text_encoder", "pytorch_model.bin"))
    own = hf.state_dict()
    if not any(k.startswith("text_model.") for k in own):
        wsd = {k[len("text_model."):]: v for k, v in wsd.items()}
    missing, unexpected = hf.load_state_dict(wsd, strict=False)
    assert not unexpected and all(k.endswith("position_ids") for k in missing)
    with torch.no_grad():
        want_text = hf(ids)[0]
    assert rel(lp["text"], want_text) < 1e-2


@pytest.mark.parametrize("kind", ["t2v", "i2v_sketch"])
def test_launcher_shards_examples_and_reproduces_the_serial_run(runs, kind):
    """motionclone_amd.launch under a 2-rank torchrun environment (gloo): rank r runs the unmodified script on lines
    r, r+2, ...; with the serial-RNG burn the sharded videos are bit-identical to the single-process run (quirk 10).
    i2v_sketch = the pixel-condition SparseCtrl: its condition images are NOT VAE-encoded, so a skipped example burns only
    the reference video's posterior draw (motionclone_functions.py:122-128)."""
    import socket
    work, rec = runs(kind)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, PYTHONPATH=ROOT, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "entry_harness.py"), kind, str(work),
                                       "--launch"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env,
                                      cwd=str(work)))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "ENTRY_OK" in o, o[-4000:]
    assert '"world": 2' in outs[0] and '"examples": 2' in outs[0]
    # checkpoints: rank 0 read the files, rank 1 read nothing from disk and received every one of them by broadcast
    import json as _json
    r1 = [_json.loads(ln) for ln in outs[1].splitlines() if ln.startswith('{"rank": 1')][0]
    r0 = [_json.loads(ln) for ln in outs[0].splitlines() if ln.startswith('{"examples"')][0]
    assert r1["checkpoint_files_read_from_disk"] == 0 and r1["received_by_broadcast"] == r0["checkpoint_files_read_by_rank0"] >= 3
    serial = [np.load(v + ".npy") for v in rec["videos"]]
    for r in range(2):
        name = os.path.basename(rec["videos"][r])
        got = np.load(os.path.join(str(work), "videos_rank%d" % r, name + ".npy"))
        assert np.array_equal(got, serial[r]), "rank %d video differs from the serial run" % r
        assert os.path.exists(os.path.join(str(work), "mr_sharded", "rank%d" % r, "clip.pt"))
        other = os.path.join(str(work), "videos_rank%d" % r, os.path.basename(rec["videos"][1 - r]) + ".npy")
        assert not os.path.exists(other)      # every example ran exactly once


def test_launcher_lanes_reproduce_the_serial_run(runs):
    """`motionclone_amd.launch --lanes 2` in ONE process: two host threads run the unmodified t2v script on alternate lines
    of the examples file, each with a private copy of the serial RNG stream (motionclone_amd/lanes.py); both videos are
    bit-identical to the single-process run and every example ran exactly once."""
    work, rec = runs("t2v")
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "entry_harness.py"), "t2v", str(work), "--launch",
                          "--lanes", "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, cwd=str(work))
    out = p.communicate(timeout=1500)[0]
    assert p.returncode == 0 and "ENTRY_OK" in out, out[-4000:]
    assert '"lanes": 2' in out and '"examples": 2' in out
    serial = [np.load(v + ".npy") for v in rec["videos"]]
    for k in range(2):
        name = os.path.basename(rec["videos"][k])
        got = np.load(os.path.join(str(work), "videos_rank0", name + ".npy"))
        assert np.array_equal(got, serial[k]), "lane %d video differs from the serial run" % k
        assert os.path.exists(os.path.join(str(work), "mr_sharded", "rank0_lane%d" % k, "clip.pt"))


def test_eight_ranks_run_the_eight_pairs_of_config_3_bit_identical_to_the_serial_run(tmp_path):
    """BASELINE config 3 without the hardware (round-4 verdict, item 9): the eight lines of the reference's
    configs/t2v_camera.jsonl:1-8 - three reference videos, seeds 42, 42, 2026, default, 2026, 2026, 2026, default
    (t2v_video_sample.py:75-105) - through `motionclone_amd.launch` under an EIGHT-rank gloo job, one line per rank, on the host
    simulator; every video is bit-identical to the single-process run of the unmodified script over all eight lines (the
    launcher burns the skipped examples' draws from the global generator, quirk 10), every line ran exactly once, and
    rank 0 alone read the checkpoint files."""
    import json as _json
    import socket
    import entry_harness as EH
    from motionclone_amd import build
    build.build_emu()          # once, here: nine children would otherwise race to rebuild a stale simulator library
    work = str(tmp_path)
    env0 = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env0.pop(k, None)
    serial = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "entry_harness.py"), "t2v", work, "--camera8"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env0, cwd=work)
    # the sharded job needs the assets the serial child writes first (checkpoints, configs, the examples file)
    import time
    t0 = time.time()
    while not os.path.exists(os.path.join(work, "examples.jsonl")):
        assert serial.poll() is None and time.time() - t0 < 600, "the serial run did not write its assets"
        time.sleep(0.5)
    time.sleep(1.0)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(8):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "entry_harness.py"), "t2v", work, "--launch"],
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, cwd=work))
    out_serial = serial.communicate(timeout=2400)[0]
    assert serial.returncode == 0 and "ENTRY_OK" in out_serial, out_serial[-4000:]
    outs = [p.communicate(timeout=2400)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "ENTRY_OK" in o, o[-4000:]
    assert '"world": 8' in outs[0] and '"examples": 8' in outs[0]
    r0 = [_json.loads(ln) for ln in outs[0].splitlines() if ln.startswith('{"examples"')][0]
    for r in range(1, 8):
        rr = [_json.loads(ln) for ln in outs[r].splitlines() if ln.startswith('{"rank": %d' % r)][0]
        assert rr["checkpoint_files_read_from_disk"] == 0 and rr["received_by_broadcast"] == r0["checkpoint_files_read_by_rank0"] >= 3
    rec = torch.load(os.path.join(work, "record.pt"))
    want = ["%s_%s%d_%d.mp4" % (stem, prompt.replace(" ", "_"), seed or 2025, seed or 2025) for stem, prompt, seed in EH.CAMERA8]
    assert [os.path.basename(v) for v in rec["videos"]] == want
    seen = set()
    for r in range(8):
        mine = sorted(f for f in os.listdir(os.path.join(work, "videos_rank%d" % r)) if f.endswith(".npy"))
        assert mine == [want[r] + ".npy"], (r, mine)          # line r -> rank r, and nothing else
        got = np.load(os.path.join(work, "videos_rank%d" % r, mine[0]))
        assert np.array_equal(got, np.load(rec["videos"][r] + ".npy")), "rank %d: video differs from the serial run" % r
        seen.add(mine[0])
    assert len(seen) == 8
    # different seeds / prompts / reference videos really gave different videos (the comparison above is not vacuous)
    vids = [np.load(v + ".npy") for v in rec["videos"]]
    assert not np.array_equal(vids[0], vids[1]) and not np.array_equal(vids[3], vids[7])
