"""Golden vectors produced by the reference itself (tests/golden/make_golden.py; the reference ships none):
  * the oracle restatement reproduces them (CPU, runs everywhere);
  * the HIP engine reproduces them within fp16 tolerance (simulator here, gfx950 on the GPU box)."""
import os

import torch

from motionclone_amd import ops
from motionclone_amd.engine import UNet3DEngine
from motionclone_amd.sampler import MotionCloneSampler
from oracle import guidance_ref as G
from oracle import unet3d_ref as U

GOLD = torch.load(os.path.join(os.path.dirname(__file__), "golden", "tiny_reference.pt"), weights_only=False)


def weights():
    sd = {k: v.half().float() for k, v in U.random_state_dict(GOLD["cfg"], seed=1234).items()}
    chk = float(sum(v.double().abs().sum() for v in sd.values()))
    assert abs(chk - GOLD["weight_checksum"]) < 1e-6 * GOLD["weight_checksum"], "seeded weights drifted"
    return sd


def close(a, b, rel):
    a, b = a.float().cpu(), b.float().cpu()
    return (a - b).norm().item() <= rel * b.norm().item()


def test_oracle_reproduces_reference_golden():
    g, cfg, sd = GOLD, GOLD["cfg"], weights()
    N, Gs, gscale = g["schedule"]
    ts = G.uneven_timesteps(N, Gs, gscale)
    assert ts.tolist() == g["timesteps"].tolist()
    with torch.no_grad():
        eps = U.unet_forward(sd, cfg, g["lat"].expand(2, -1, -1, -1, -1), int(ts[0]), g["text"])
    assert close(eps, g["eps_b2"], 1e-5)
    rep = G.extract_representation(sd, cfg, g["vid"], g["noise"], g["text"][[0]])
    for k, (v, i) in g["rep"].items():
        assert torch.equal(rep[k][1], i) and close(rep[k][0], v, 1e-5)
    hp = dict(g["hp"], guidance_steps=Gs)
    nxt, aux = G.guided_step(sd, cfg, g["lat"], 0, ts, g["text"], g["rep"], hp)
    assert close(aux["grad"], g["guided_score"], 1e-4) and close(nxt, g["guided_next"], 1e-5)
    p, _ = G.plain_step_full(sd, cfg, g["guided_next"], Gs, ts, g["text"], g["hp"]["cfg_scale"])
    assert close(p, g["plain_next"], 1e-5)
    last, _ = G.plain_step_full(sd, cfg, g["plain_next"], N - 1, ts, g["text"], g["hp"]["cfg_scale"])
    assert close(last, g["last_next"], 1e-5)


def test_engine_reproduces_reference_golden(backend):
    dev = backend
    g, cfg, sd = GOLD, GOLD["cfg"], weights()
    N, Gs, gscale = g["schedule"]
    eng = UNet3DEngine(sd, cfg, dev)
    smp = MotionCloneSampler(eng, num_inference_steps=N, guidance_steps=Gs, guidance_scale=gscale, **g["hp"])
    lat, text = g["lat"].half().to(dev), g["text"].half().to(dev)
    eps = eng.forward(lat.expand(2, -1, -1, -1, -1), int(smp.timesteps[0]), text)
    assert close(ops.cl_to_latent(eps, 2, 4, 4, 8, 8), g["eps_b2"], 2e-2)
    rep = smp.extract(g["vid"].half().to(dev), g["noise"].half().to(dev), text[0:1])
    # index work: the uint8 arg-max must EQUAL the reference's, except where the fp32 probabilities (recomputed by the oracle,
    # which the test above pins to this same golden file) hold a tie at the 5e-4 level - measured: 0 of 192 rows differ
    rec = {}
    with torch.no_grad():
        noisy = G.add_noise(G.alphas_cumprod(), 400, g["vid"], g["noise"])
        U.unet_forward(sd, cfg, noisy.float(), 400, g["text"][[0]], only_motion_feature=True, record=rec)
        prob = G.temp_attn_prob(rec, cfg["motion_heads"])
    flips = total = 0
    for k, (v, i) in g["rep"].items():
        assert close(rep[k][0], v, 1e-2)
        mine = rep[k][1].cpu()
        mism = mine != i
        flips, total = flips + int(mism.sum()), total + mism.numel()
        if mism.any():
            gap = (prob[k].max(-1, keepdim=True).values - torch.gather(prob[k], -1, mine.long()))[mism]
            assert float(gap.max()) <= 5e-4, "an arg-max flip that is not a tie: gap %g" % float(gap.max())
    assert flips <= max(1, total // 100), (flips, total)
    rep_dev = eng.prepare_representation(g["rep"])
    aux = {}
    nxt = smp.step(lat, 0, text, rep_dev, aux=aux)
    assert close(aux["grad"], g["guided_score"], 5e-2)
    assert close(nxt, g["guided_next"], 2e-2)
    p = smp.step(g["guided_next"].half().to(dev), Gs, text, rep_dev)
    assert close(p, g["plain_next"], 2e-2)
    last = smp.step(g["plain_next"].half().to(dev), N - 1, text, rep_dev)
    assert close(last, g["last_next"], 2e-2)
