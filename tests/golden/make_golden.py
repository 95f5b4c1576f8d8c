"""Generates tests/golden/tiny_reference.pt by running the UNMODIFIED reference code (imported from
/root/reference through oracle/reference_shim.py) on the tiny UNet3D with seeded weights and inputs.

Run in the authoring container:  python tests/golden/make_golden.py
The fixture holds inputs, a weight checksum and the reference's outputs: B=2 forward, motion representation
(top-1 values + uint8 indices of the 6 hooked temporal attentions), one guided single_step_video (next latents,
eps_c+cfg mix, guidance gradient as passed to customized_step), one plain step and the last step."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import reference_shim as shim  # noqa: E402
from oracle import unet3d_ref as U  # noqa: E402

HP = dict(cfg_scale=7.5, motion_guidance_weight=2000.0, warm_up_steps=10, cool_up_steps=10)
N, G, GSCALE = 4, 2, 0.3


def inputs(cfg, F=4, H=8, W=8, n_text=7):
    lat = torch.randn(1, 4, F, H, W, generator=torch.Generator().manual_seed(2025)).half().float()
    text = torch.randn(2, n_text, cfg["cross_attention_dim"], generator=torch.Generator().manual_seed(7)).half().float()
    vid = (0.18215 * torch.randn(1, 4, F, H, W, generator=torch.Generator().manual_seed(11))).half().float()
    noise = torch.randn(1, 4, F, H, W, generator=torch.Generator().manual_seed(2025)).half().float()
    return lat, text, vid, noise


def main():
    cfg = dict(U.TINY_CONFIG)
    sd = {k: v.half().float() for k, v in U.random_state_dict(cfg, seed=1234).items()}
    lat, text, vid, noise = inputs(cfg)
    H = shim.RefHarness(cfg, sd, HP, N, G, GSCALE)
    captured = {}
    orig = H.sched.customized_step

    def spy(model_output, step_index, sample, score=None, **kw):
        captured["noise_pred"] = model_output.detach().clone()
        captured["score"] = None if score is None else score.detach().clone()
        return orig(model_output, step_index, sample, score=score, **kw)
    H.sched.customized_step = spy
    out = dict(cfg=cfg, hp=HP, schedule=(N, G, GSCALE), lat=lat, text=text, vid=vid, noise=noise,
               weight_checksum=float(sum(v.double().abs().sum() for v in sd.values())),
               timesteps=H.sched.timesteps.clone())
    with torch.no_grad():
        out["eps_b2"] = H.unet(lat.expand(2, -1, -1, -1, -1), int(H.sched.timesteps[0]), encoder_hidden_states=text).sample
    rep = H.extract(vid, noise, text[[0]])
    out["rep"] = {k: [v[0].clone(), v[1].clone()] for k, v in rep.items()}
    nxt = H.step(lat, 0, text, rep)
    out["guided_next"], out["guided_noise_pred"], out["guided_score"] = nxt.clone(), captured["noise_pred"], captured["score"]
    p = H.step(nxt, G, text, rep)
    out["plain_next"], out["plain_noise_pred"] = p.clone(), captured["noise_pred"]
    out["last_next"] = H.step(p, N - 1, text, rep).clone()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiny_reference.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
