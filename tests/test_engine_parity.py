"""End-to-end parity of the HIP engine with the oracle on the tiny UNet3D: forward, motion-representation
extraction, guidance loss + latent gradient, one guided and one plain DDIM step, a short loop.

'emu' runs the kernel sources on the host simulator here; 'hip' (gpu-marked) runs the real library."""
import pytest
import torch

from motionclone_amd.engine import UNet3DEngine
from motionclone_amd.sampler import MotionCloneSampler
from motionclone_amd import ops
from oracle import guidance_ref as G
from oracle import unet3d_ref as U

HP = dict(cfg_scale=7.5, motion_guidance_weight=2000.0, warm_up_steps=10, cool_up_steps=10)


def make_inputs(cfg, F=4, H=8, W=8, n_text=7):
    lat = torch.randn(1, 4, F, H, W, generator=torch.Generator().manual_seed(2025))
    text = torch.randn(2, n_text, cfg["cross_attention_dim"], generator=torch.Generator().manual_seed(7))
    vid = 0.18215 * torch.randn(1, 4, F, H, W, generator=torch.Generator().manual_seed(11))
    noise = torch.randn(1, 4, F, H, W, generator=torch.Generator().manual_seed(2025))
    return lat, text, vid, noise


def rel_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def to_lat(eps_tokens, B, F, H, W):
    return ops.cl_to_latent(eps_tokens, B, 4, F, H, W).float().cpu()


@pytest.fixture
def tiny():
    cfg = dict(U.TINY_CONFIG)
    sd = U.random_state_dict(cfg, seed=1234)
    # weights rounded to fp16 once, so oracle (fp32 math) and engine (fp16 storage) share identical parameters
    sd = {k: v.half().float() for k, v in sd.items()}
    return cfg, sd


def test_forward_matches_oracle(backend, tiny):
    dev = backend
    cfg, sd = tiny
    lat, text, _, _ = make_inputs(cfg)
    lat16, text16 = lat.half(), text.half()
    eng = UNet3DEngine(sd, cfg, dev)
    eps = eng.forward(lat16.expand(2, -1, -1, -1, -1).to(dev), 701, text16.to(dev))
    with torch.no_grad():
        ref = U.unet_forward(sd, cfg, lat16.float().expand(2, -1, -1, -1, -1), 701, text16.float())
    got = to_lat(eps, 2, 4, 8, 8)
    assert rel_err(got, ref) < 2e-2, rel_err(got, ref)


def test_extraction_matches_oracle(backend, tiny):
    dev = backend
    cfg, sd = tiny
    _, text, vid, noise = make_inputs(cfg)
    eng = UNet3DEngine(sd, cfg, dev)
    smp = MotionCloneSampler(eng, num_inference_steps=4, guidance_steps=2, guidance_scale=0.3, **HP)
    rep = smp.extract(vid.half().to(dev), noise.half().to(dev), text[[0]].half().to(dev))
    noisy = smp.add_noise(400, vid.half(), noise.half()).float()
    rec = {}
    with torch.no_grad():
        U.unet_forward(sd, cfg, noisy, 400, text[[0]].half().float(), only_motion_feature=True, record=rec)
        prob = G.temp_attn_prob(rec, cfg["motion_heads"])
    ref = G.motion_representation(prob)
    assert list(rep) == list(ref)
    for k in ref:
        v, i = rep[k]
        assert v.shape == ref[k][0].shape and i.dtype == torch.uint8
        assert (v.float().cpu() - ref[k][0]).abs().max() < 5e-3
        mism = i.cpu() != ref[k][1]
        if mism.any():  # index flips only where the two best probabilities are within fp16 noise
            p = prob[k]
            alt = torch.gather(p, -1, i.cpu().long())
            assert ((ref[k][0] - alt)[mism] < 5e-3).all()
            assert mism.float().mean() < 0.02


@pytest.mark.parametrize("batch_guided", [True, False], ids=["batchedB2", "twoB1"])
def test_guided_and_plain_step_match_oracle(backend, tiny, batch_guided):
    dev = backend
    cfg, sd = tiny
    lat, text, vid, noise = make_inputs(cfg)
    lat16, text16 = lat.half(), text.half()
    N, Gs, gscale = 4, 2, 0.3
    hp = dict(HP, guidance_steps=Gs)
    rep = G.extract_representation(sd, cfg, vid, noise, text16[[0]].float())
    ts = G.uneven_timesteps(N, Gs, gscale)
    eng = UNet3DEngine(sd, cfg, dev)
    smp = MotionCloneSampler(eng, num_inference_steps=N, guidance_steps=Gs, guidance_scale=gscale,
                             batch_guided=batch_guided, **HP)
    assert smp.timesteps.tolist() == ts.tolist()
    rep_dev = eng.prepare_representation(rep)

    aux = {}
    nxt = smp.step(lat16.to(dev), 0, text16.to(dev), rep_dev, aux=aux)
    ref_nxt, ref_aux = G.guided_step(sd, cfg, lat16.float(), 0, ts, text16.float(), rep, hp)
    assert rel_err(to_lat(aux["eps_c"], 1, 4, 8, 8), ref_aux["eps_c"]) < 2e-2
    assert rel_err(to_lat(aux["eps_u"], 1, 4, 8, 8), ref_aux["eps_u"]) < 2e-2
    assert abs(aux["loss"].item() - ref_aux["loss"].item()) < 3e-2 * abs(ref_aux["loss"].item())
    g_err = rel_err(aux["grad"], ref_aux["grad"])
    assert g_err < 5e-2, g_err
    assert rel_err(nxt, ref_nxt) < 2e-2

    aux2 = {}
    p = smp.step(nxt, Gs, text16.to(dev), rep_dev, aux=aux2)
    ref_p, _ = G.plain_step_full(sd, cfg, nxt.float().cpu(), Gs, ts, text16.float(), HP["cfg_scale"])
    assert rel_err(p, ref_p) < 2e-2
    last = smp.step(p, N - 1, text16.to(dev), rep_dev)
    ref_last, _ = G.plain_step_full(sd, cfg, p.float().cpu(), N - 1, ts, text16.float(), HP["cfg_scale"])
    assert rel_err(last, ref_last) < 2e-2


def test_forward_more_than_16_frames(backend, tiny):
    """F = 20 -> two 16-wide tiles per side in the temporal-attention kernels (the config-5 code path, F = 32)"""
    dev = backend
    cfg, sd = tiny
    lat = torch.randn(1, 4, 20, 8, 8, generator=torch.Generator().manual_seed(5)).half()
    text = torch.randn(1, 7, cfg["cross_attention_dim"], generator=torch.Generator().manual_seed(7)).half()
    eng = UNet3DEngine(sd, cfg, dev)
    eps = eng.forward(lat.to(dev), 301, text.to(dev))
    with torch.no_grad():
        ref = U.unet_forward(sd, cfg, lat.float(), 301, text.float())
    assert rel_err(ops.cl_to_latent(eps, 1, 4, 20, 8, 8).float().cpu(), ref) < 2e-2


def test_multiple_guidance_blocks_match_oracle(backend, tiny):
    """motion_guidance_blocks with more than one entry: every temporal attention whose name contains one of them is hooked
    (util.py:434-440), the index of the LAST entry bounds the differentiated half (motionclone_functions.py:602)"""
    dev = backend
    cfg, sd = tiny
    lat, text, vid, noise = make_inputs(cfg)
    blocks = ["down_blocks.2", "up_blocks.1"]
    eng = UNet3DEngine(sd, cfg, dev, guidance_blocks=blocks)
    names = eng.hooked_names()
    assert len(names) == 4 + 6 and names[0].startswith("down_blocks.2.motion_modules.0.")
    smp = MotionCloneSampler(eng, num_inference_steps=4, guidance_steps=2, guidance_scale=0.3, **HP)
    rep = smp.extract(vid.half().to(dev), noise.half().to(dev), text[[0]].half().to(dev))
    noisy = smp.add_noise(400, vid.half(), noise.half()).float()
    rec = {}
    with torch.no_grad():
        U.unet_forward(sd, cfg, noisy, 400, text[[0]].half().float(), only_motion_feature=True, record=rec, hooked=tuple(blocks))
        ref = G.motion_representation(G.temp_attn_prob(rec, cfg["motion_heads"]))
    assert list(rep) == list(ref) == names
    for k in ref:
        assert (rep[k][0].float().cpu() - ref[k][0]).abs().max() < 5e-3
    # gradient of the 10-module loss
    lat16, text16 = lat.half(), text.half()
    t = int(smp.timesteps[0])
    _, grad, loss = eng.guided_eps_and_grad(lat16.to(dev), t, text16[1:2].to(dev), eng.prepare_representation(ref), 2000.0,
                                            want_loss=True)
    control = lat16.float().clone().requires_grad_(True)
    rec = {}
    U.unet_forward(sd, cfg, control, t, text16[[1]].float(), record=rec, hooked=tuple(blocks))
    ref_loss = 2000.0 * G.temp_loss(G.temp_attn_prob(rec, cfg["motion_heads"]), ref)
    (ref_grad,) = torch.autograd.grad(ref_loss, control)
    assert abs(float(loss) - float(ref_loss)) < 3e-2 * abs(float(ref_loss))
    assert rel_err(grad, ref_grad) < 5e-2
    with pytest.raises(NotImplementedError):
        UNet3DEngine(sd, cfg, dev, guidance_blocks=["up_blocks.2", "up_blocks.1"])


def test_interleaved_sampling_equals_one_at_a_time(backend, tiny):
    """sampler.sample_interleaved: step i of every job before step i + 1 of any (own sampler per lane, shared engine);
    on the GPU each lane has its own HIP stream, on the host simulator the lanes share the (only) stream.  Results must be
    bit-identical to running the jobs one after the other, and the GEMM share hint must not change them beyond rounding."""
    from motionclone_amd.sampler import sample_interleaved
    dev = backend
    cfg, sd = tiny
    eng = UNet3DEngine(sd, cfg, dev)
    lat, text, vid, noise = [t.half().to(dev) for t in make_inputs(cfg, F=2 if dev.type == "cpu" else 4)]
    lat2 = torch.randn(lat.shape, generator=torch.Generator().manual_seed(99)).half().to(dev)
    N = 1 if dev.type == "cpu" else 3      # the host simulator runs a single (guided) step per job: issue order and plumbing
    jobs = [(lat, text, vid, noise), (lat2, text, vid.flip(2).contiguous(), noise)]

    def mk():
        return MotionCloneSampler(eng, num_inference_steps=N, guidance_steps=max(1, N - 1), guidance_scale=0.4, **HP)
    seq = []
    for la, tx, vd, nz in jobs:
        s = mk()
        seq.append(s.sample(la, tx, s.extract(vd, nz, tx[0:1])).clone())
    streams = [torch.cuda.Stream(device=dev) for _ in jobs] if dev.type == "cuda" else None
    order = []
    out = sample_interleaved([mk(), mk()], jobs, streams, on_step=lambda k, i, enter: order.append((i, k)) if enter else None)
    if streams is not None:
        torch.cuda.synchronize()
    assert order == [(i, k) for i in range(N) for k in range(2)]
    assert all(torch.equal(a, b) for a, b in zip(out, seq))
    assert sample_interleaved([mk()], [], streams) == []
    with pytest.raises(ValueError, match="2 jobs for 1 samplers"):
        sample_interleaved([mk()], jobs, streams)
    if dev.type == "cuda":
        try:
            ops.set_gemm_share(2)
            shared = sample_interleaved([mk(), mk()], jobs, streams)
            torch.cuda.synchronize()
        finally:
            ops.set_gemm_share(1)
        assert all(rel_err(a, b) < 5e-3 for a, b in zip(shared, seq))


def test_step_with_eta_matches_the_scheduler_formula(backend, tiny):
    """extra_step_kwargs of single_step_video (motionclone_functions.py:241,255): eta > 0 changes the direction coefficient
    to sqrt(1 - a_prev - sigma^2) and adds sigma * noise (:364-365,386,391-405); a guided and a plain step"""
    dev = backend
    cfg, sd = tiny
    eng = UNet3DEngine(sd, cfg, dev)
    lat, text, vid, noise = [t.half().to(dev) for t in make_inputs(cfg, F=2)]
    N, Gs, gscale = 4, 2, 0.3        # step Gs is a plain step that is not the last one (the last has sigma = 0)
    smp = MotionCloneSampler(eng, num_inference_steps=N, guidance_steps=Gs, guidance_scale=gscale, **HP)
    rep_dev = eng.prepare_representation(smp.extract(vid, noise, text[0:1]))
    ts = G.uneven_timesteps(N, Gs, gscale)
    B, _, F, H, W = lat.shape
    z = torch.randn(lat.shape, generator=torch.Generator().manual_seed(21)).half()
    for i in (0, Gs):
        aux = {}
        plain0 = smp.step(lat, i, text, rep_dev, aux=aux)             # eta = 0: hands out eps_u, eps_c (and the gradient)
        got = smp.step(lat, i, text, rep_dev, eta=0.6, variance_noise=z.to(dev))
        eu, ec = to_lat(aux["eps_u"], 1, F, H, W), to_lat(aux["eps_c"], 1, F, H, W)
        eps = ec + HP["cfg_scale"] * (ec - eu)
        want = G.ddim_step_general(G.alphas_cumprod(), torch.tensor(G.FINAL_ALPHA_CUMPROD), ts, i, eps, lat.float().cpu(),
                                   eta=0.6, variance_noise=z.float(), score=aux["grad"].float().cpu() if i < Gs else None,
                                   guidance_scale=smp.score_gs)[0]
        assert rel_err(got, want) < 5e-3
        assert rel_err(got, plain0) > 1e-3    # and it is not the eta = 0 update (sigma_t is small late in the schedule)
    i = Gs
    drawn = smp.step(lat, i, text, rep_dev, eta=0.6, generator=torch.Generator(device=dev).manual_seed(21))
    z2 = torch.randn(lat.shape, generator=torch.Generator(device=dev).manual_seed(21), device=dev, dtype=lat.dtype)
    assert torch.equal(drawn, smp.step(lat, i, text, rep_dev, eta=0.6, variance_noise=z2))
    with pytest.raises(ValueError, match="Cannot pass both generator and variance_noise"):
        smp.step(lat, i, text, rep_dev, eta=0.6, variance_noise=z2, generator=torch.Generator(device=dev))


def test_videos_batched_in_one_launch_sequence_match_their_separate_steps(backend, tiny):
    """round 4: V independent videos through ONE launch sequence (latents [V, ...], text [u_1 .. u_V | c_1 .. c_V], the tape
    differentiating the conditional halves): every video's guided step (eps, guidance gradient with ITS OWN loss mean),
    plain step and the interleaved sampler loop agree with its separate V = 1 run and with the oracle"""
    from motionclone_amd.sampler import sample_interleaved
    dev = backend
    cfg, sd = tiny
    eng = UNet3DEngine(sd, cfg, dev)
    N, Gs, gs = 3, 2, 0.3
    smp = MotionCloneSampler(eng, num_inference_steps=N, guidance_steps=Gs, guidance_scale=gs, **HP)
    vids = []
    for v in range(2):
        g = torch.Generator().manual_seed(100 + v)
        lat = torch.randn(1, 4, 4, 8, 8, generator=g).half().to(dev)
        text = torch.randn(2, 7, cfg["cross_attention_dim"], generator=g).half().to(dev)
        vid = (0.18215 * torch.randn(1, 4, 4, 8, 8, generator=g)).half().to(dev)
        noise = torch.randn(1, 4, 4, 8, 8, generator=g).half().to(dev)
        vids.append((lat, text, vid, noise))
    reps = [smp.extract(vid, noise, text[0:1]) for (_, text, vid, noise) in vids]
    rep_devs = [eng.prepare_representation(r) for r in reps]
    rep_cat = eng.prepare_representation(reps)
    lat2 = torch.cat([v[0] for v in vids], 0)
    text2 = torch.cat([v[1][0:1] for v in vids] + [v[1][1:2] for v in vids], 0)
    ts = G.uneven_timesteps(N, Gs, gs)
    for i in (0, Gs):                      # one guided, one plain step
        aux2 = {}
        nxt2 = smp.step(lat2, i, text2, rep_cat, aux=aux2)
        assert nxt2.shape == lat2.shape
        l_sep = 0.0
        for v, (lat, text, _, _) in enumerate(vids):
            aux1 = {}
            nxt1 = smp.step(lat, i, text, rep_devs[v], aux=aux1)
            if i < Gs:
                l_sep += float(aux1["loss"])
            assert rel_err(nxt2[v:v + 1], nxt1) < 2e-3, (i, v, rel_err(nxt2[v:v + 1], nxt1))
            if i < Gs:
                assert rel_err(aux2["grad"][v:v + 1], aux1["grad"]) < 2e-2
                if v == 1:      # and against the oracle (the V = 1 path is held to it by the tests above)
                    rep_cpu = {k: [a.float().cpu(), b.cpu()] for k, (a, b) in reps[v].items()}
                    ref, ref_aux = G.guided_step(sd, cfg, lat.float().cpu(), i, ts, text.float().cpu(), rep_cpu, dict(HP, guidance_steps=Gs))
                    assert rel_err(nxt2[v:v + 1], ref) < 2e-2 and rel_err(aux2["grad"][v:v + 1], ref_aux["grad"]) < 5e-2
        if i < Gs:   # the loss of the batch is the sum of the videos' own means (already computed by the steps above)
            assert abs(float(aux2["loss"]) - l_sep) < 2e-3 * abs(l_sep), (float(aux2["loss"]), l_sep)
    # the sampler loop (a short schedule): both videos as ONE batched lane vs one lane each
    s1 = MotionCloneSampler(eng, num_inference_steps=2, guidance_steps=1, guidance_scale=gs, **HP)
    s2 = MotionCloneSampler(eng, num_inference_steps=2, guidance_steps=1, guidance_scale=gs, **HP)
    a = sample_interleaved([s1], [list(vids)])[0]
    b = sample_interleaved([s1, s2], vids)
    assert a.shape[0] == 2
    for v in range(2):
        assert rel_err(a[v:v + 1], b[v]) < 5e-3, rel_err(a[v:v + 1], b[v])


@pytest.mark.parametrize("V", [1, 2])
def test_shared_prefix_of_the_cfg_batch_equals_the_duplicated_batch(backend, tiny, V):
    """forward(dup=True) (round 5): the CFG batch [u_1 .. u_V | c_1 .. c_V] holds the same latents twice
    (motionclone_functions.py:216-223,248-253), so conv_in, the first ResnetBlock3D and the text-free part of the first
    transformer run ONCE on V elements.  Same eps as the duplicated batch (to the fp32 summation order of GEMMs whose tile
    choice follows the row count), same guidance gradient, and both against the oracle; the engine's A/B switch
    `share_prefix` selects the old launch sequence."""
    dev = backend
    cfg, sd = tiny
    lat, text, vid, noise = make_inputs(cfg)
    lats = torch.cat([lat + 0.1 * v for v in range(V)], 0).half()
    tu = torch.cat([text[[0]] + 0.05 * v for v in range(V)], 0).half()
    tc = torch.cat([text[[1]] - 0.05 * v for v in range(V)], 0).half()
    text2 = torch.cat([tu, tc], 0)
    eng = UNet3DEngine(sd, cfg, dev)
    shared = to_lat(eng.forward(lats.to(dev), 701, text2.to(dev), dup=True), 2 * V, 4, 8, 8)
    full = to_lat(eng.forward(torch.cat([lats, lats], 0).to(dev), 701, text2.to(dev)), 2 * V, 4, 8, 8)
    with torch.no_grad():
        ref = U.unet_forward(sd, cfg, torch.cat([lats, lats], 0).float(), 701, text2.float())
    assert rel_err(shared, full) < 4e-3, rel_err(shared, full)
    assert rel_err(shared, ref) < 2e-2 and rel_err(full, ref) < 2e-2
    with pytest.raises(ValueError):
        eng.forward(lats.to(dev), 701, text2[:V].to(dev), dup=True)
    # the taped (guided) path: gradient of the conditional halves through the junction
    reps = [G.extract_representation(sd, cfg, vid + 0.01 * v, noise, tu[[v]].float()) for v in range(V)]
    rep_dev = eng.prepare_representation(reps if V > 1 else reps[0])
    ec1, g1, l1, eu1 = eng.guided_eps_and_grad(lats.to(dev), 701, tc.to(dev), rep_dev, 2000.0, want_loss=True, text_uncond=tu.to(dev))
    if V == 1:      # (one A/B on the host simulator is enough: the switch only selects the launch sequence)
        eng.share_prefix = False
        ec0, g0, l0, eu0 = eng.guided_eps_and_grad(lats.to(dev), 701, tc.to(dev), rep_dev, 2000.0, want_loss=True, text_uncond=tu.to(dev))
        eng.share_prefix = True
        assert rel_err(g1, g0) < 2e-2, rel_err(g1, g0)
        assert rel_err(ec1.float(), ec0.float()) < 4e-3 and rel_err(eu1.float(), eu0.float()) < 4e-3
        assert abs(float(l1) - float(l0)) < 2e-3 * abs(float(l0))
    hp = dict(HP, guidance_steps=2)
    for v in range(V):      # each video against its own oracle step (the loss weight is applied un-scaled: step factor 1)
        control = lats[[v]].float().clone().requires_grad_(True)
        rec = {}
        U.unet_forward(sd, cfg, control, 701, tc[[v]].float(), record=rec, hooked=("up_blocks.1",))
        loss = 2000.0 * G.temp_loss(G.temp_attn_prob(rec, cfg["motion_heads"]), reps[v])
        (gref,) = torch.autograd.grad(loss, control)
        assert rel_err(g1[v:v + 1], gref) < 5e-2, rel_err(g1[v:v + 1], gref)
