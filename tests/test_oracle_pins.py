"""Pins the oracle restatement (oracle/*.py) to the reference's own, unmodified code.

Runs only where /root/reference exists (the authoring container); the GPU box relies on the golden
fixtures these same calls produced (tests/golden/, tests/test_golden.py)."""
import pytest
import torch

from oracle import guidance_ref as G
from oracle import reference_shim as shim
from oracle import unet3d_ref as U

pytestmark = pytest.mark.skipif(not shim.available(), reason="reference tree not present")

def _close(a, b, rel=2e-5):
    return (a - b).abs().max().item() <= rel * max(1.0, b.abs().max().item())


HP = dict(cfg_scale=7.5, motion_guidance_weight=2000.0, warm_up_steps=10, cool_up_steps=10)


def _inputs(cfg, F=4, H=8, W=8):
    g = torch.Generator().manual_seed(2025)
    lat = torch.randn(1, 4, F, H, W, generator=g)
    text = torch.randn(2, 7, cfg["cross_attention_dim"], generator=torch.Generator().manual_seed(7))
    vid = 0.18215 * torch.randn(1, 4, F, H, W, generator=torch.Generator().manual_seed(11))
    noise = torch.randn(1, 4, F, H, W, generator=torch.Generator().manual_seed(2025))
    return lat, text, vid, noise


def test_state_dict_inventory_matches_reference():
    for cfg in (U.TINY_CONFIG, U.SD15_CONFIG):
        with torch.device("meta"):
            ref = shim.reference_unet(cfg)
        ref_shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items() if "pos_encoder" not in k}
        mine = {k: tuple(v) for k, v in U.param_shapes(cfg).items()}
        assert set(ref_shapes) == set(mine)
        for k in mine:
            assert ref_shapes[k] == mine[k], k
    n = sum(torch.Size(s).numel() for s in U.param_shapes(U.SD15_CONFIG).values())
    assert abs(n / 1e6 - 1276.7) < 0.2  # SURVEY.md 2b [probe]


def test_unet_forward_and_steps_match_reference():
    cfg = U.TINY_CONFIG
    sd = U.random_state_dict(cfg, seed=1234)
    lat, text, vid, noise = _inputs(cfg)
    N, Gs, gscale = 4, 2, 0.3
    H = shim.RefHarness(cfg, sd, HP, N, Gs, gscale)
    ts = G.uneven_timesteps(N, Gs, gscale)
    assert ts.tolist() == H.sched.timesteps.tolist()
    assert torch.allclose(G.alphas_cumprod(), H.sched.alphas_cumprod)

    # plain forward, B = 2
    with torch.no_grad():
        ref = H.unet(lat.expand(2, -1, -1, -1, -1), int(ts[0]), encoder_hidden_states=text).sample
        mine = U.unet_forward(sd, cfg, lat.expand(2, -1, -1, -1, -1), int(ts[0]), text)
    assert torch.allclose(ref, mine, atol=2e-5, rtol=1e-4), (ref - mine).abs().max()

    # motion representation extraction
    rep_ref = H.extract(vid, noise, text[[0]])
    rep = G.extract_representation(sd, cfg, vid, noise, text[[0]])
    assert list(rep_ref) == list(rep) and len(rep) == 6
    for k in rep:
        assert torch.equal(rep_ref[k][1], rep[k][1]), k
        assert torch.allclose(rep_ref[k][0], rep[k][0], atol=1e-6), k

    # one guided and one plain step of single_step_video
    hp = dict(HP, guidance_steps=Gs)
    nxt_ref = H.step(lat, 0, text, rep_ref)
    nxt, aux = G.guided_step(sd, cfg, lat, 0, ts, text, rep, hp)
    assert aux["grad"].abs().max() > 0
    assert _close(nxt, nxt_ref), (nxt_ref - nxt).abs().max()
    p_ref = H.step(nxt_ref, Gs, text, rep_ref)
    p, _ = G.plain_step_full(sd, cfg, nxt, Gs, ts, text, HP["cfg_scale"])
    assert _close(p, p_ref), (p_ref - p).abs().max()
    # last step uses final_alpha_cumprod = 1
    l_ref = H.step(p_ref, N - 1, text, rep_ref)
    l, _ = G.plain_step_full(sd, cfg, p, N - 1, ts, text, HP["cfg_scale"])
    assert _close(l, l_ref), (l_ref - l).abs().max()


def test_chunked_attention_of_the_oracle_equals_the_unchunked_path(monkeypatch):
    """the frame-chunked attention with recompute-in-backward (what lets the fp32 oracle differentiate 32 f x 96 x 96) against
    the plain `_attention` restatement that the test above pins to the reference: forward and q / k / v gradients"""
    g = torch.Generator().manual_seed(3)
    B, N, M, heads, d = 5, 24, 17, 2, 8
    q0, k0, v0 = (torch.randn(B, n, heads * d, generator=g, dtype=torch.float64) for n in (N, M, M))
    w = torch.randn(B, N, heads * d, generator=g, dtype=torch.float64)
    res = []
    for limit in (U.MHA_MAX_SCORE_BYTES, 2 * heads * N * M * 8):       # unchunked; chunks of 2 batch entries (last one ragged)
        monkeypatch.setattr(U, "MHA_MAX_SCORE_BYTES", limit)
        q, k, v = (t.clone().requires_grad_() for t in (q0, k0, v0))
        out = U._mha(q, k, v, heads)
        res.append((out.detach(),) + torch.autograd.grad((out * w).sum(), (q, k, v)))
        with torch.no_grad():
            assert torch.equal(U._mha(q0, k0, v0, heads), out.detach())
    for a, b in zip(*res):
        assert torch.allclose(a, b, rtol=1e-12, atol=1e-13), (a - b).abs().max()
    assert torch.equal(res[0][0], res[1][0])     # the forward is the same arithmetic slice by slice
