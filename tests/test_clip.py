"""CLIP text encoder against the real third-party implementation: `transformers.CLIPTextModel`
is installed in this image, so the HIP engine is compared with it directly (tiny config on the simulator / GPU, the
SD-1.5 text-tower size on the GPU), plus the kernel pieces it adds (causal attention, embedding lookup, quick_gelu)."""
import pytest
import torch

from motionclone_amd import ops
from motionclone_amd.clip_engine import ClipTextEngine

transformers = pytest.importorskip("transformers")

TINY = dict(vocab_size=97, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
            max_position_embeddings=77, layer_norm_eps=1e-5, hidden_act="quick_gelu")


def hf_model(cfg, seed):
    torch.manual_seed(seed)
    hc = transformers.CLIPTextConfig(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"],
                                     intermediate_size=cfg["intermediate_size"], num_hidden_layers=cfg["num_hidden_layers"],
                                     num_attention_heads=cfg["num_attention_heads"],
                                     max_position_embeddings=cfg["max_position_embeddings"], hidden_act="quick_gelu",
                                     layer_norm_eps=cfg["layer_norm_eps"], attn_implementation="eager")
    m = transformers.CLIPTextModel(hc).eval()
    with torch.no_grad():
        for p in m.parameters():          # HF init is tiny (std 0.02 * ...): widen it so that every op matters
            if p.dim() > 1:
                p.mul_(4.0)
            p.copy_(p.half().float())
    return m


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def test_causal_attention_embed_quick_gelu(backend):
    dev = backend
    g = torch.Generator().manual_seed(0)
    nb, N, H, d = 2, 77, 2, 32
    C = H * d
    qkv = (torch.randn(nb * N, 3 * C, generator=g) * 0.7).half().to(dev)
    o = ops.attn_fwd_causal(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], N, H, d, nb)
    t = qkv.float().cpu().reshape(nb, N, 3, H, d).permute(2, 0, 3, 1, 4)
    s = (t[0] @ t[1].transpose(-1, -2)) * d ** -0.5
    s = s.masked_fill(torch.triu(torch.ones(N, N, dtype=torch.bool), 1), float("-inf"))
    ref = (s.softmax(-1) @ t[2]).permute(0, 2, 1, 3).reshape(nb * N, C)
    assert (o.float().cpu() - ref).abs().max() < 1e-2
    ids = torch.randint(0, 50, (2, 9), generator=g)
    tok, pos = torch.randn(50, 64, generator=g).half(), torch.randn(77, 64, generator=g).half()
    e = ops.clip_embed(ids.to(dev), tok.to(dev), pos.to(dev))
    assert torch.allclose(e.float().cpu(), (tok[ids].float() + pos[:9].float()).reshape(18, 64), atol=2e-3)
    x = (torch.randn(16, 64, generator=g) * 3).half()
    assert torch.allclose(ops.quick_gelu(x.to(dev)).float().cpu(), x.float() * torch.sigmoid(1.702 * x.float()), atol=4e-3)


def test_text_encoder_matches_transformers(backend):
    dev = backend
    m = hf_model(TINY, 0)
    eng = ClipTextEngine(m.state_dict(), TINY, dev)
    ids = torch.randint(0, TINY["vocab_size"], (2, 77), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = m(ids)[0]
    got = eng.forward(ids.to(dev))
    assert got.shape == (2, 77, 64) and rel(got, ref) < 1e-2, rel(got, ref)
    short = eng.forward(ids[:, :20].contiguous().to(dev))          # causal: a prefix sees only itself
    assert rel(short, ref[:, :20]) < 1e-2


def test_dropin_wrapper_and_encode_prompt(backend):
    dev = backend
    from motionclone_amd.models.clip import CLIPTextModel, clip_param_shapes
    m = hf_model(TINY, 2)
    te = CLIPTextModel(**TINY)
    assert set(te.state_dict()) == set(clip_param_shapes(TINY))
    assert all(k.startswith("text_model.") for k in te.state_dict())              # the 4.28.1 / SD checkpoint layout
    sd = dict(m.state_dict())                                                     # (this transformers may omit the prefix)
    pre = "text_model." if any(k.startswith("text_model.") for k in sd) else ""
    sd[pre + "embeddings.position_ids"] = torch.arange(77).unsqueeze(0)           # old-checkpoint buffer: ignored
    te.load_state_dict({k: (v.half() if v.is_floating_point() else v) for k, v in sd.items()})
    te = te.to(dev)
    ids = torch.randint(0, TINY["vocab_size"], (1, 77), generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        ref = m(ids)[0]
    out = te(ids.to(dev), attention_mask=None)
    assert isinstance(out, tuple) and rel(out[0], ref) < 1e-2
    assert te.config.max_position_embeddings == 77 and te.dtype == torch.float16


@pytest.mark.gpu
def test_sd15_text_tower_matches_transformers():
    from motionclone_amd import lib
    from motionclone_amd.clip_engine import SD15_CLIP_CONFIG
    lib._lib = None
    lib._is_emulated = False
    lib.load()
    dev = torch.device("cuda:0")
    m = hf_model(SD15_CLIP_CONFIG, 4)
    eng = ClipTextEngine(m.state_dict(), SD15_CLIP_CONFIG, dev)
    ids = torch.randint(0, 49408, (2, 77), generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        ref = m(ids)[0]
    got = eng.forward(ids.to(dev))
    assert got.shape == (2, 77, 768) and rel(got, ref) < 1e-2, rel(got, ref)
    assert torch.equal(got, eng.forward(ids.to(dev)))
