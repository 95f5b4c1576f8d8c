"""Parity at the sizes that are benchmarked: SD-1.5 widths (320/640/1280, head dims 40/80/160) + AnimateDiff motion
modules at BASELINE's shapes, HIP engine vs the oracle run in fp32 ON THE SAME MI355X (the oracle is plain torch; its
convolutions use PyTorch's im2col + GEMM path, see parity_util.oracle_mode).

  config 1 shape  16 f x 32 x 32 latent (256^2),  schedule (10, 5, 0.3)  - forward, extraction, guided, plain, full loop
  config 2 shape  16 f x 64 x 64 latent (512^2),  schedule (30, 18, 0.4) - forward, extraction, guided, plain, last step
  config 4        config 2 + SparseCtrl (i2v_rgb), schedule (30, 12, 0.3) - encoder residuals, extraction, guided, plain
  config 5        32 f x 96 x 96 latent (768^2),  schedule (50, 30, 0.4) - B = 1 forward (+ the fp16 oracle stays finite),
                  extraction, guided step + backward, plain and last step (round 4: the oracle's chunked attention)
  second witness  the oracle in fp16 on the GPU (= the reference's own arithmetic through stock PyTorch-ROCm)

  config 2 loop        ALL 30 steps of the (30, 18, 0.4) schedule (round 3: steps 14 .. 21)
  config 4 trajectory  steps 8 .. 15 of (30, 12, 0.3) with the SparseCtrl encoder on every step
  F = 32 (config 5)    32 f x 32 x 32 and 32 f x 48 x 48 latents: forward, extraction, guided (two-tile temporal backward,
                       guidance seed at (F = 32, d = 160)), plain and last step

Weights: synthetic seed 1234 with motion proj_out re-randomised; both sides use the same fp16-rounded
parameters.  Tolerances (parity_util, 3-4x the measured errors): forward / latents 5e-3 relative L2, loss 0.2 %, arg-max flips
must be ties (oracle gap <= 5e-4), at most 0.5 % of the rows, and are counted exactly; round 5: guidance gradient 1.2e-2
(1.5e-2 at config 5's real size) = 2x the measured errors.  Measured errors are written to gpurun_out/parity_r06.json.

Round 5 additions: the fp16 oracle (the reference's arithmetic through stock PyTorch) as second witness for the guidance
LOSS, GRADIENT and the whole 30-step config-2 trajectory (reported next to the engine's distance from the fp32 oracle);
config 4 over ALL 30 steps; the top-1 / probability kernels against the reference's own fp16 torch ops on the GPU, fed with
the q / k the engine recorded at config 2 and config 5 size (bit-equal away from fp16 rounding boundaries).
"""
import pytest
import torch

import parity_util as PU
from motionclone_amd import spec
from motionclone_amd.engine import ControlNetEngine, UNet3DEngine, default_config
from motionclone_amd.sampler import MotionCloneSampler
from oracle import guidance_ref as G
from oracle import unet3d_ref as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world():
    from motionclone_amd import lib
    lib._lib = None
    lib._is_emulated = False
    lib.load()
    dev = torch.device("cuda:0")
    cfg = default_config()
    assert {k: cfg[k] for k in U.SD15_CONFIG} == U.SD15_CONFIG
    sd, _ = spec.synthetic_state_dict(cfg, seed=1234, device=dev)
    eng = UNet3DEngine(sd, cfg, dev)
    sdo = PU.oracle_weights(sd, dev)
    return dev, cfg, sd, eng, sdo


def sampler(eng, N, Gs, gs, controlnet=None):
    return MotionCloneSampler(eng, num_inference_steps=N, guidance_steps=Gs, guidance_scale=gs, controlnet=controlnet, **PU.HP)


@pytest.mark.parametrize("shape,sched", [((16, 32, 32), (10, 5, 0.3)), ((16, 64, 64), (30, 18, 0.4))], ids=["cfg1", "cfg2"])
def test_forward_extraction_guided_plain(world, shape, sched):
    dev, cfg, sd, eng, sdo = world
    F, H, W = shape
    key = "cfg1_16f_256" if H == 32 else "cfg2_16f_512"
    lat, text, vid, noise = PU.synth_inputs(cfg, F, H, W, dev)
    smp = sampler(eng, *sched)
    PU.check_forward_b2(eng, sdo, cfg, lat, text, int(smp.timesteps[0]), key)
    _, rep_ref, _ = PU.check_extraction(eng, smp, sdo, cfg, vid, noise, text, key)
    nxt, _ = PU.check_guided_step(eng, smp, sdo, cfg, lat, text, rep_ref, 0, key, tol_grad=PU.TOL_GRAD_FULLSIZE,
                                  witness_sd16=PU.fp16_weights(sd))
    nxt2, _ = PU.check_plain_step(eng, smp, sdo, cfg, nxt, text, smp.G, key)
    PU.check_plain_step(eng, smp, sdo, cfg, nxt2, text, smp.N - 1, key)      # last step: alpha_prev = final_alpha_cumprod
    torch.cuda.empty_cache()


def test_full_loop_config1(world):
    """the whole config-1 schedule (10 steps, 5 guided): engine and oracle each follow their own trajectory"""
    dev, cfg, sd, eng, sdo = world
    F, H, W = 16, 32, 32
    lat, text, vid, noise = PU.synth_inputs(cfg, F, H, W, dev)
    smp = sampler(eng, 10, 5, 0.3)
    with PU.oracle_mode(dev):
        rep_ref = G.extract_representation(sdo, cfg, vid.float(), noise.float(), text[0:1].float())
    PU.check_loop(eng, smp, sdo, cfg, lat, text, rep_ref, "cfg1_16f_256", tol=6e-3)
    torch.cuda.empty_cache()


def test_full_loop_config2(world):
    """BASELINE config 2 end to end: ALL 30 steps of the (30, 18, 0.4) schedule - 18 guided (warm-up scaling on steps 0 .. 9,
    cool-down on 9 .. 17), the switch, 12 plain, the last step with alpha_prev = final_alpha_cumprod - engine and fp32 oracle
    each on their own trajectory from the seeded initial latent (round 3 covered steps 14 .. 21 only)."""
    dev, cfg, sd, eng, sdo = world
    F, H, W = 16, 64, 64
    lat, text, vid, noise = PU.synth_inputs(cfg, F, H, W, dev)
    smp = sampler(eng, 30, 18, 0.4)
    with PU.oracle_mode(dev):
        rep_ref = G.extract_representation(sdo, cfg, vid.float(), noise.float(), text[0:1].float())
    # measured 6.0e-4 -> 1.5e-3; round 5: the reference's own fp16 arithmetic follows as a third trajectory (witness)
    PU.check_loop(eng, smp, sdo, cfg, lat, text, rep_ref, "cfg2_full_loop", tol=5e-3, witness_sd16=PU.fp16_weights(sd))
    torch.cuda.empty_cache()


def test_full_loop_config4_sparsectrl(world):
    """BASELINE config 4 (i2v_rgb + SparseCtrl, schedule (30, 12, 0.3)): ALL 30 steps (round 4: steps 8 .. 15) - 12 guided
    with warm-up and cool-down, the switch, 18 plain, the last step - the SparseCtrl encoder re-run on every step's timestep
    on both sides, engine and oracle each on their own trajectory from the seeded initial latent."""
    dev, cfg, sd, eng, sdo = world
    F, H, W = 16, 64, 64
    csd = spec.synthetic_controlnet_state_dict(cfg, seed=4321, device=dev)
    ceng = ControlNetEngine(csd, cfg, dev)
    csdo = PU.oracle_weights(csd, dev)
    lat, text, vid, noise = PU.synth_inputs(cfg, F, H, W, dev)
    smp = sampler(eng, 30, 12, 0.3, controlnet=ceng)
    cond = torch.zeros_like(vid)
    mask = torch.zeros_like(vid[:, :1])
    cond[:, :, 0] = vid[:, :, 0]
    mask[:, :, 0] = 1
    ctrl = dict(cond=cond, mask=mask, scale=1.0)
    noisy = smp.add_noise(400, vid, noise).float()
    with torch.no_grad(), PU.oracle_mode(dev):
        dr, mr = U.controlnet_forward(csdo, cfg, noisy.shape, 400, text[0:1].float(), cond.float(), mask.float(), 1.0)
        rec = {}
        U.unet_forward(sdo, cfg, noisy, 400, text[0:1].float(), only_motion_feature=True, record=rec, down_residuals=dr,
                       mid_residual=mr)
        rep_ref = G.motion_representation(G.temp_attn_prob(rec, cfg["motion_heads"]))
    del dr, mr, rec
    PU.check_loop(eng, smp, sdo, cfg, lat, text, rep_ref, "cfg4_full_loop", tol=5e-3, ctrl=ctrl, csdo=csdo)
    torch.cuda.empty_cache()


@pytest.mark.parametrize("hw", [32, 48])
def test_32_frames_forward_extraction_guided_plain(world, hw):
    """BASELINE config 5 has F = 32 (the PE table's maximum, motion_module.py:60): temporal attention runs as two 16-frame
    tiles per side, the backward's F > 16 path, the guidance seed and top-1 at (F = 32, d = 160), the tape at that size.
    Spatial size reduced to what the fp32 oracle's autograd fits comfortably; schedule (50, 30, 0.4) as config 5."""
    dev, cfg, sd, eng, sdo = world
    F, H, W = 32, hw, hw
    key = "cfg5_32f_%d" % (8 * hw)
    lat, text, vid, noise = PU.synth_inputs(cfg, F, H, W, dev)
    smp = sampler(eng, 50, 30, 0.4)
    PU.check_forward_b2(eng, sdo, cfg, lat, text, int(smp.timesteps[0]), key)
    _, rep_ref, _ = PU.check_extraction(eng, smp, sdo, cfg, vid, noise, text, key)
    nxt, _ = PU.check_guided_step(eng, smp, sdo, cfg, lat, text, rep_ref, 0, key, tol_grad=PU.TOL_GRAD_FULLSIZE)
    nxt2, _ = PU.check_plain_step(eng, smp, sdo, cfg, nxt, text, smp.G, key)
    PU.check_plain_step(eng, smp, sdo, cfg, nxt2, text, smp.N - 1, key)
    torch.cuda.empty_cache()


def test_config5_32_frames_768_forward_extraction_guided_plain(world):
    """BASELINE config 5 at its REAL size: 32 f x 96 x 96 latent (level-0 self-attention over 9216 tokens, two 16-frame
    temporal tiles, the PE table's last row), schedule (50, 30, 0.4): B = 1 forward, extraction, the guided step with its
    backward, a plain and the last step.  The fp32 oracle differentiates this size through its frame-chunked attention with
    recompute (oracle/unet3d_ref.py:_ChunkedAttention; the unchunked restatement would keep 2 x 87 GB of probabilities).
    The oracle in fp16 (= the reference's own arithmetic) must stay finite here too: the saturating conversions of the
    kernels (mc_common.hpp to_half) never act where the reference would have produced inf."""
    dev, cfg, sd, eng, sdo = world
    F, H, W = 32, 96, 96
    key = "cfg5_32f_768"
    lat, text, vid, noise = PU.synth_inputs(cfg, F, H, W, dev)
    smp = sampler(eng, 50, 30, 0.4)
    t0 = int(smp.timesteps[0])
    eps = eng.forward(lat, t0, text[1:2])
    got = PU.to_lat(eps, 1, F, H, W)
    del eps
    torch.cuda.empty_cache()
    with torch.no_grad(), PU.oracle_mode(dev):
        ref = U.unet_forward(sdo, cfg, lat.float(), t0, text[1:2].float())
        sd16 = {k: v.half() for k, v in sd.items()}
        ref16 = U.unet_forward(sd16, cfg, lat, t0, text[1:2])
        del sd16
    e = PU.rel(got, ref)
    assert torch.isfinite(ref16).all(), "the reference's fp16 arithmetic overflows at config 5"
    PU.report(key, forward_b1_rel=e, witness_fp16_oracle_forward_rel=PU.rel(ref16, ref), fp16_oracle_finite=True)
    assert e < PU.TOL_FWD, e
    del got, ref, ref16
    torch.cuda.empty_cache()
    _, rep_ref, _ = PU.check_extraction(eng, smp, sdo, cfg, vid, noise, text, key)
    torch.cuda.empty_cache()
    nxt, _ = PU.check_guided_step(eng, smp, sdo, cfg, lat, text, rep_ref, 0, key, tol_grad=PU.TOL_GRAD_CONFIG5)
    torch.cuda.empty_cache()
    nxt2, _ = PU.check_plain_step(eng, smp, sdo, cfg, nxt, text, smp.G, key)
    PU.check_plain_step(eng, smp, sdo, cfg, nxt2, text, smp.N - 1, key)
    torch.cuda.empty_cache()


def test_sparsectrl_config4(world):
    """i2v_rgb: SparseCtrl encoder residuals, extraction / guided / plain step with them (motionclone_functions.py:176-208)"""
    dev, cfg, sd, eng, sdo = world
    F, H, W = 16, 64, 64
    key = "cfg4_sparsectrl"
    csd = spec.synthetic_controlnet_state_dict(cfg, seed=4321, device=dev)
    ceng = ControlNetEngine(csd, cfg, dev)
    csdo = PU.oracle_weights(csd, dev)
    lat, text, vid, noise = PU.synth_inputs(cfg, F, H, W, dev)
    smp = sampler(eng, 30, 12, 0.3, controlnet=ceng)
    cond = torch.zeros_like(vid)
    mask = torch.zeros_like(vid[:, :1])
    cond[:, :, 0] = vid[:, :, 0]
    mask[:, :, 0] = 1
    scale = 1.0
    ctrl = dict(cond=cond, mask=mask, scale=scale)
    t0 = int(smp.timesteps[0])

    def tok(t):
        return t.permute(0, 2, 3, 4, 1).reshape(-1, t.shape[1])
    down, mid = ceng.forward((2, 4, F, H, W), t0, text, cond, mask, scale)
    with torch.no_grad(), PU.oracle_mode(dev):
        d_ref, m_ref = U.controlnet_forward(csdo, cfg, (2, 4, F, H, W), t0, text.float(), cond.float(), mask.float(), scale)
    errs = [PU.rel(a, tok(b)) for a, b in zip(down, d_ref)] + [PU.rel(mid, tok(m_ref))]
    PU.report(key, controlnet_residual_rel_max=max(errs))
    assert len(down) == 12 and max(errs) < PU.TOL_FWD, errs
    del down, mid
    # extraction with the encoder (uncond text, t = 400)
    noisy = smp.add_noise(400, vid, noise).float()
    with torch.no_grad(), PU.oracle_mode(dev):
        dr, mr = U.controlnet_forward(csdo, cfg, noisy.shape, 400, text[0:1].float(), cond.float(), mask.float(), scale)
    _, rep_ref, _ = PU.check_extraction(eng, smp, sdo, cfg, vid, noise, text, key, ctrl=ctrl, res=(dr, mr))
    res_u = ([d[[0]] for d in d_ref], m_ref[[0]])
    res_c = ([d[[1]] for d in d_ref], m_ref[[1]])
    nxt, _ = PU.check_guided_step(eng, smp, sdo, cfg, lat, text, rep_ref, 0, key, ctrl=ctrl, res_u=res_u, res_c=res_c,
                                  tol_grad=PU.TOL_GRAD_FULLSIZE)
    tG = int(smp.timesteps[smp.G])
    with torch.no_grad(), PU.oracle_mode(dev):
        d3, m3 = U.controlnet_forward(csdo, cfg, (2, 4, F, H, W), tG, text.float(), cond.float(), mask.float(), scale)
    PU.check_plain_step(eng, smp, sdo, cfg, nxt, text, smp.G, key, ctrl=ctrl, res=(d3, m3))
    torch.cuda.empty_cache()


def test_fp16_oracle_second_witness(world):
    """the oracle in fp16 on the GPU is the reference's own arithmetic through stock PyTorch-ROCm.  Its
    distance to the fp32 oracle is the 'fp16 tolerance' of the claim; the engine (fp16 storage, fp32 accumulation) must
    not be further away than the tolerance, and is reported next to it.  A finite fp16 forward also shows that no
    activation of the reference reaches the fp16 range limit at this size, i.e. the saturating conversions of the
    kernels (mc_common.hpp to_half) never act where the reference would have produced inf."""
    dev, cfg, sd, eng, sdo = world
    F, H, W = 16, 64, 64
    key = "cfg2_16f_512"
    lat, text, vid, noise = PU.synth_inputs(cfg, F, H, W, dev)
    smp = sampler(eng, 30, 18, 0.4)
    t = int(smp.timesteps[0])
    sd16 = {k: v.half() for k, v in sd.items()}
    with torch.no_grad(), PU.oracle_mode(dev):
        ref32 = U.unet_forward(sdo, cfg, lat.float().expand(2, -1, -1, -1, -1), t, text.float())
        ref16 = U.unet_forward(sd16, cfg, lat.expand(2, -1, -1, -1, -1), t, text)
    assert torch.isfinite(ref16).all(), "the reference's fp16 arithmetic overflows at this size"
    got = PU.to_lat(eng.forward(lat.expand(2, -1, -1, -1, -1), t, text), 2, F, H, W)
    e16, eng_e = PU.rel(ref16, ref32), PU.rel(got, ref32)
    PU.report(key, witness_fp16_oracle_forward_rel=e16, witness_engine_forward_rel=eng_e,
              witness_engine_vs_fp16_oracle=PU.rel(got, ref16))
    assert eng_e < PU.TOL_FWD
    # extraction in fp16: how many arg-max indices the reference's own arithmetic flips against fp32
    noisy = smp.add_noise(400, vid, noise)
    rec32, rec16 = {}, {}
    with torch.no_grad(), PU.oracle_mode(dev):
        U.unet_forward(sdo, cfg, noisy.float(), 400, text[0:1].float(), only_motion_feature=True, record=rec32)
        U.unet_forward(sd16, cfg, noisy, 400, text[0:1], only_motion_feature=True, record=rec16)
        p32 = G.temp_attn_prob(rec32, cfg["motion_heads"])
        rep16 = G.motion_representation(G.temp_attn_prob(rec16, cfg["motion_heads"]))
    flips = total = 0
    for k in p32:
        n, tot, gap, _ = PU.flip_stats(rep16[k][1], rep16[k][0], p32[k])
        flips, total = flips + n, total + tot
    PU.report(key, witness_fp16_oracle_extraction_flips=flips, witness_rows=total)
    torch.cuda.empty_cache()


@pytest.mark.parametrize("shape", [(16, 64, 64), (32, 96, 96)], ids=["cfg2", "cfg5"])
def test_top1_and_prob_equal_the_reference_fp16_torch_ops_on_the_engines_own_qk(world, shape):
    """Index work end to end (round-4 verdict, weak 2): the q / k the ENGINE records in its extraction forward at config 2 and
    config 5 size go (a) through mc_tattn_top1_f16 / mc_tattn_prob_f16 and (b) through the reference's own fp16 torch ops ON
    THE GPU - reshape_heads_to_batch_dim (attention.py:367-372), get_attention_scores = baddbmm(beta 0, alpha scale) ->
    softmax(-1) -> .to(fp16) (attention.py:564-611), reshape (motionclone_functions.py:279), topk(k=1) + uint8 (:79).
    The uint8 indices, the fp16 top values and the whole fp16 probability tensor must be EQUAL, except on rows an exact (fp64)
    evaluation marks ambiguous: a score within the fp32 dot-product error of an fp16 rounding boundary, or a probability
    within 2e-6 relative of one - there the reference's own result depends on its GEMM's summation order.  Counted and
    reported; fewer than 1 % of the rows may be ambiguous-and-different (measured 0.28 % at config 2, 0.58 % at config 5), and
    on those rows the two fp16 probability tensors stay within one score rounding of each other (measured <= 0.18 % relative,
    bound 0.5 %); the uint8 INDICES were equal on every row of both sizes."""
    dev, cfg, sd, eng, sdo = world
    F, H, W = shape
    key = "cfg2_16f_512" if F == 16 else "cfg5_32f_768"
    lat, text, vid, noise = PU.synth_inputs(cfg, F, H, W, dev)
    smp = sampler(eng, 30, 18, 0.4)
    noisy = smp.add_noise(400, vid, noise)
    record = {}
    eng.forward(noisy, 400, text[0:1], record=record, only_motion_feature=True)
    from motionclone_amd import ops
    rows = differ = amb_rows = unexplained = ties = idx_differs = 0
    worst_rel = 0.0
    for name in eng.hooked_names():
        r = record[name]
        C, g, heads, d = r["C"], r["geo"], r["heads"], r["d"]
        q_tok, k_tok = r["qkv"][:, :C], r["qkv"][:, C:2 * C]
        val, idx = ops.tattn_top1(q_tok, k_tok, g.B, g.F, g.hw, heads, d)
        prob = ops.tattn_prob(q_tok, k_tok, g.B, g.F, g.hw, heads, d)

        def seq(t):      # token matrix [(b f n), C] -> the processor's [(b n), f, C] (motion_module.py:279)
            return t.reshape(g.B, g.F, g.hw, C).permute(0, 2, 1, 3).reshape(g.B * g.hw, g.F, C).contiguous()

        def heads_to_batch(t):   # attention.py:367-372
            b, s_, dim = t.shape
            return t.reshape(b, s_, heads, dim // heads).permute(0, 2, 1, 3).reshape(b * heads, s_, dim // heads).contiguous()
        qh, kh = heads_to_batch(seq(q_tok)), heads_to_batch(seq(k_tok))
        scale = d ** -0.5
        scores = torch.baddbmm(torch.empty(qh.shape[0], qh.shape[1], kh.shape[1], dtype=qh.dtype, device=dev), qh,
                               kh.transpose(-1, -2), beta=0, alpha=scale)
        p_ref = scores.softmax(dim=-1).to(qh.dtype).reshape(-1, heads, g.F, g.F)
        v_ref, i_ref = torch.topk(p_ref, k=1, dim=-1)
        i_ref = i_ref.to(torch.uint8)
        # exact evaluation of the reference's order of operations
        q64, k64 = qh.double(), kh.double()
        sc = float(torch.tensor(scale, dtype=torch.float32))
        s64 = (q64 @ k64.transpose(-1, -2)) * sc
        mag = (q64.abs() @ k64.abs().transpose(-1, -2)) * sc

        def near(x, tol):
            return (x + tol).half() != (x - tol).half()
        p64 = torch.softmax(s64.half().double(), dim=-1)
        amb = (near(s64, 1e-6 * mag) | near(p64, 2e-6 * p64)).any(-1).reshape(-1, heads, g.F)
        # equal fp16 maxima in a row: the kernel takes the lowest index (what the CPU topk does); the device topk may return any
        # of them, so on such rows the kernel's index only has to point at one of the maxima
        tie = (p_ref == v_ref).sum(-1) > 1
        at_idx = torch.gather(p_ref, -1, idx.long())
        idx_ok = (idx[..., 0] == i_ref[..., 0]) | (tie & (at_idx[..., 0] == v_ref[..., 0]))
        bad = ~idx_ok | (val[..., 0] != v_ref[..., 0]) | (prob != p_ref).any(-1)
        ties += int(tie.sum())
        idx_differs += int((~idx_ok).sum())
        # where the fp16 tensors are not bit-equal they are one score-rounding apart: an fp16 score off by one ulp (<= 2^-11 of
        # |s| <= ~8) moves a probability by <= ~0.4 %
        dp = ((prob.float() - p_ref.float()).abs() / p_ref.float().clamp_min(1e-4)).amax(-1)
        worst_rel = max(worst_rel, float(dp[bad].max()) if bad.any() else 0.0)
        rows += bad.numel()
        differ += int(bad.sum())
        amb_rows += int(amb.sum())
        unexplained += int((bad & ~amb).sum())
        del scores, p_ref, s64, mag, p64, q64, k64
    PU.report(key, top1_vs_reference_fp16_torch_rows=rows, top1_vs_reference_fp16_torch_rows_differing=differ,
              top1_rows_on_a_rounding_boundary=amb_rows, top1_rows_differing_off_boundary=unexplained,
              top1_rows_with_equal_fp16_maxima=ties, top1_rows_with_another_index=idx_differs,
              top1_differing_rows_max_rel_prob_difference=worst_rel)
    assert unexplained == 0, "%d rows differ from the reference's fp16 torch path away from any rounding boundary" % unexplained
    assert differ <= 0.01 * rows, (differ, rows)
    assert worst_rel < 5e-3, worst_rel          # measured 1.2e-3 (config 2) / 1.8e-3 (config 5): one or two fp16 ulps
    assert idx_differs <= 1e-4 * rows, (idx_differs, rows)   # measured: 0 of 196608 and 0 of 884736 - the uint8 indices are EQUAL
    torch.cuda.empty_cache()


@pytest.mark.parametrize("variant", ["outlier_channels", "sharp_attention", "bos_token"])
def test_outlier_channels_stress(world, variant):
    """All other full-size cases run on N(0, init) weights; trained SD-1.5 / AnimateDiff weights are known for a few channels that
    carry activations tens of times larger than the rest.  No checkpoint exists offline, so the statistics are provoked
    instead: 4 channels of every GroupNorm / LayerNorm gain x 12 and two output channels of every FeedForward x 6 (seeded), at
    config 1's size.  The engine (fp16 storage, fp32 accumulation, gradients carried at grad_scale) must stay as close to the
    fp32 oracle as on the plain weights - forward, extraction ties, loss, guidance gradient, latents - and the reference's own
    fp16 arithmetic is run next to it as the witness of how much of the deviation is fp16 itself.

    Variant "sharp_attention": every to_q / to_k (spatial self, cross and temporal) x 2.5, so the logits of every attention span
    6 x what N(0, init) gives (rows outgrow their first key tile at every level, the temporal maps turn nearly one-hot).
    Variant "bos_token": the first text token x 20 - the norm ratio of CLIP's BOS embedding to the rest - on both prompts.
    (Both at once make the random network ill-conditioned: the reference's own fp16 arithmetic is then 0.17 from fp32, and so
    is the engine - measured, not kept as a test.)  All bounds are max(the usual tolerance, 1.25 x the fp16 witness's own
    distance from the fp32 oracle)."""
    dev, cfg, sd, eng0, sdo0 = world
    sd2, n = PU.stress_weights(sd, variant, dev)
    assert {"outlier_channels": n["norm"] > 100 and n["ff"] == 36, "sharp_attention": n["qk"] == 144, "bos_token": True}[variant]
    eng = UNet3DEngine(sd2, cfg, dev)
    sdo = PU.oracle_weights(sd2, dev)
    F, H, W = 16, 32, 32
    key = {"outlier_channels": "cfg1_outlier_stress", "sharp_attention": "cfg1_sharp_attention_stress", "bos_token": "cfg1_bos_token_stress"}[variant]
    lat, text, vid, noise = PU.synth_inputs(cfg, F, H, W, dev)
    if variant == "bos_token":
        text = text.clone()
        text[:, 0] *= 20.0
    smp = sampler(eng, 10, 5, 0.3)
    t0 = int(smp.timesteps[0])
    got = PU.to_lat(eng.forward(lat, t0, text, dup=True), 2, F, H, W)
    with torch.no_grad(), PU.oracle_mode(dev):
        ref = U.unet_forward(sdo, cfg, lat.float().expand(2, -1, -1, -1, -1), t0, text.float())
        ref16 = U.unet_forward(PU.fp16_weights(sd2), cfg, lat.expand(2, -1, -1, -1, -1), t0, text)
    e_fwd, w_fwd = PU.rel(got, ref), PU.rel(ref16, ref)
    PU.report(key, forward_b2_rel=e_fwd, eps_abs_max=ref.abs().max(), witness_fp16_oracle_forward_rel=w_fwd,
              witness_fp16_oracle_finite=bool(torch.isfinite(ref16).all()), eps_abs_max_plain_weights=1.67)
    assert torch.isfinite(got).all() and e_fwd < max(PU.TOL_FWD, 1.25 * w_fwd), (e_fwd, w_fwd)
    del got, ref, ref16
    # extraction: arg-max flips of the engine AND of the reference's fp16 arithmetic against the fp32 oracle's probabilities (sharper
    # maps and larger scores widen what an fp16 score rounding can flip: the bound is the witness's own figure, not TIE_GAP)
    rep = smp.extract(vid, noise, text[0:1], add_noise_step=400)
    noisy = smp.add_noise(400, vid, noise)
    rec32, rec16 = {}, {}
    with torch.no_grad(), PU.oracle_mode(dev):
        U.unet_forward(sdo, cfg, noisy.float(), 400, text[0:1].float(), only_motion_feature=True, record=rec32)
        U.unet_forward(PU.fp16_weights(sd2), cfg, noisy, 400, text[0:1], only_motion_feature=True, record=rec16)
        p32 = G.temp_attn_prob(rec32, cfg["motion_heads"])
        rep16 = G.motion_representation(G.temp_attn_prob(rec16, cfg["motion_heads"]))
    rep_ref = G.motion_representation(p32)
    fl = tot = wfl = 0
    gap = wgap = 0.0
    for k_ in p32:
        n, t_, g_, _ = PU.flip_stats(rep[k_][1], rep[k_][0], p32[k_])
        wn, _, wg, _ = PU.flip_stats(rep16[k_][1], rep16[k_][0], p32[k_])
        fl, tot, wfl, gap, wgap = fl + n, tot + t_, wfl + wn, max(gap, g_), max(wgap, wg)
    PU.report(key, extraction_flips=fl, extraction_rows=tot, extraction_flip_max_gap=gap, witness_fp16_oracle_extraction_flips=wfl,
              witness_fp16_oracle_extraction_flip_max_gap=wgap)
    assert fl <= 1.5 * wfl + 0.001 * tot and gap <= max(PU.TIE_GAP, 1.5 * wgap), (fl, wfl, gap, wgap)
    del rec32, rec16, p32, rep16
    nxt, _ = PU.check_guided_step(eng, smp, sdo, cfg, lat, text, rep_ref, 0, key, tol_grad=PU.TOL_GRAD,
                                  witness_sd16=PU.fp16_weights(sd2), witness_factor=1.25)
    # The plain step's CFG combination eps_c + 7.5 (eps_c - eps_u) amplifies whatever error the two halves do not share; with
    # these weights the text moves eps by only ~8 % (|eps_c - eps_u| / |eps_c|), so fp16 rounding of EITHER implementation shows
    # up ~100x in the latents.  "Within fp16 tolerance" is therefore measured against the second witness here: the engine may
    # not be further from the fp32 oracle than the reference's own fp16 arithmetic is (x 1.25), at three plain steps.
    ts = G.uneven_timesteps(smp.N, smp.G, smp.guidance_scale)
    sd16 = PU.fp16_weights(sd2)
    worst = 0.0
    for i in (smp.G, smp.G + 2, smp.N - 1):
        got = smp.step(nxt, i, text, {})
        with PU.oracle_mode(dev):
            ref_nxt, ref_aux = G.plain_step_full(sdo, cfg, nxt.float(), i, ts, text.float(), PU.HP["cfg_scale"])
            wit_nxt, _ = G.plain_step_full(sd16, cfg, nxt, i, ts, text, PU.HP["cfg_scale"])
        e, ew = PU.rel(got, ref_nxt), PU.rel(wit_nxt, ref_nxt)
        d = ref_aux["eps_c"] - ref_aux["eps_u"]
        PU.report(key, **{"plain_step_%d_latents" % i: e, "witness_fp16_oracle_plain_step_%d_latents" % i: ew,
                          "plain_step_%d_cfg_difference_over_eps" % i: float(d.norm() / ref_aux["eps_c"].norm())})
        assert torch.isfinite(got.float()).all()
        assert e < max(PU.TOL_FWD, 1.25 * ew), (i, e, ew)
        worst = max(worst, e / max(ew, 1e-12))
    PU.report(key, plain_steps_engine_error_over_fp16_reference_error_max=worst)
    torch.cuda.empty_cache()


def test_non_square_frames(world):
    """The reference takes any --H / --W (t2v_video_sample.py:118-120); every BASELINE config is square.  320 x 512 pixels = a 40 x 64
    latent (levels 2560 / 640 / 160 / 40 tokens: the level-0 attention on the ring kernel with a ragged last key tile... none of the
    level sizes is a power of two): forward, extraction, guided and plain step against the oracle, usual tolerances."""
    dev, cfg, sd, eng, sdo = world
    F, H, W = 16, 40, 64
    key = "nonsquare_16f_320x512"
    lat, text, vid, noise = PU.synth_inputs(cfg, F, H, W, dev)
    smp = sampler(eng, 30, 18, 0.4)
    PU.check_forward_b2(eng, sdo, cfg, lat, text, int(smp.timesteps[0]), key)
    _, rep_ref, _ = PU.check_extraction(eng, smp, sdo, cfg, vid, noise, text, key)
    nxt, _ = PU.check_guided_step(eng, smp, sdo, cfg, lat, text, rep_ref, 0, key, tol_grad=PU.TOL_GRAD_FULLSIZE)
    PU.check_plain_step(eng, smp, sdo, cfg, nxt, text, smp.G, key)
    torch.cuda.empty_cache()


@pytest.mark.parametrize("F", [8, 24])
def test_other_frame_counts(world, F):
    """--L other than 16 / 32 (the PE table allows up to 32 frames, motion_module.py:60): 8 frames (half a temporal tile) and 24
    (one and a half), 32 x 32 latents: forward, extraction, guided and plain step against the oracle."""
    dev, cfg, sd, eng, sdo = world
    H = W = 32
    key = "frames_%d_256" % F
    lat, text, vid, noise = PU.synth_inputs(cfg, F, H, W, dev)
    smp = sampler(eng, 10, 5, 0.3)
    PU.check_forward_b2(eng, sdo, cfg, lat, text, int(smp.timesteps[0]), key)
    _, rep_ref, _ = PU.check_extraction(eng, smp, sdo, cfg, vid, noise, text, key)
    nxt, _ = PU.check_guided_step(eng, smp, sdo, cfg, lat, text, rep_ref, 0, key, tol_grad=PU.TOL_GRAD_FULLSIZE)
    PU.check_plain_step(eng, smp, sdo, cfg, nxt, text, smp.G, key)
    torch.cuda.empty_cache()


# ---- round 6 (verdict item 3): guidance that MATTERS ------------------------------------------------------------------------
# On the N(0, init) weights the MotionClone score moves the latents by ~1e-5 relative (profiles/r05_parity_fullsize.json:
# guided_grad_abs_max 4e-5 at config 2 against |eps| ~ 1.6), so the 30-step drift of the loops above is, in effect, a CFG-DDIM
# drift: a factor-2 error of the gradient path would pass them.  With every to_q / to_k x 2.5 ("sharp_attention") the temporal
# maps are nearly one-hot, the loss is ~30 and the score is percents of eps: here the guidance term is tested AS A TERM.
@pytest.fixture(scope="module")
def sharp_world(world):
    dev, cfg, sd, _, _ = world
    sd2, n = PU.stress_weights(sd, "sharp_attention", dev)
    assert n["qk"] == 144
    return dev, cfg, sd2, UNet3DEngine(sd2, cfg, dev), PU.oracle_weights(sd2, dev)


def test_guidance_sensitive_loop_config2(sharp_world):
    """BASELINE config 2 (16 f x 64 x 64 latent, schedule (30, 18, 0.4)), sharp-attention weights: ALL 18 guided steps and the
    first plain steps with four trajectories - engine, fp32 oracle, oracle with the gradient HALVED, oracle WITHOUT guidance.  The
    engine must stay within half of the halved-gradient trajectory's distance (parity_util.check_guidance_sensitive_loop): a
    gradient path that is off by 2x fails this test."""
    dev, cfg, sd2, eng, sdo = sharp_world
    F, H, W = 16, 64, 64
    lat, text, vid, noise = PU.synth_inputs(cfg, F, H, W, dev)
    smp = sampler(eng, 30, 18, 0.4)
    with PU.oracle_mode(dev):
        rep_ref = G.extract_representation(sdo, cfg, vid.float(), noise.float(), text[0:1].float())
    PU.check_guidance_sensitive_loop(eng, smp, sdo, cfg, lat, text, rep_ref, "cfg2_sharp_attention_sensitive_loop", last=21)
    torch.cuda.empty_cache()


def test_guidance_sensitive_step_config5(sharp_world):
    """BASELINE config 5's real size (32 f x 96 x 96 latent, schedule (50, 30, 0.4)), sharp-attention weights: ONE guided step at
    the top of the warm-up ramp (step 9, factor 1.0) vs the fp32 oracle, next to the same step with the gradient halved / dropped;
    the gradient bound is the full-size one although the operands are the ill-conditioned ones."""
    dev, cfg, sd2, eng, sdo = sharp_world
    F, H, W = 32, 96, 96
    key = "cfg5_sharp_attention_guided_step"
    lat, text, vid, noise = PU.synth_inputs(cfg, F, H, W, dev)
    smp = sampler(eng, 50, 30, 0.4)
    with PU.oracle_mode(dev):
        rep_ref = G.extract_representation(sdo, cfg, vid.float(), noise.float(), text[0:1].float())
    ts = G.uneven_timesteps(smp.N, smp.G, smp.guidance_scale)
    hp = dict(PU.HP, guidance_steps=smp.G)
    i = 9
    aux = {}
    nxt = smp.step(lat, i, text, eng.prepare_representation(rep_ref), aux=aux)
    with PU.oracle_mode(dev):
        ref, raux = PU.oracle_guided_step_scaled(sdo, cfg, lat.float(), i, ts, text.float(), rep_ref, hp, 1.0)
        eps = raux["eps_c"] + hp["cfg_scale"] * (raux["eps_c"] - raux["eps_u"])
        acp = G.alphas_cumprod()
        half = G.ddim_step(acp, ts, i, eps, lat.float(), score=0.5 * raux["grad"])
        none = G.ddim_step(acp, ts, i, eps, lat.float(), score=None)
    e = dict(latents=PU.rel(nxt, ref), grad=PU.rel(aux["grad"], raux["grad"]),
             loss=abs(float(aux["loss"]) - float(raux["loss"])) / abs(float(raux["loss"])),
             half_gradient_latents=PU.rel(half, ref), no_guidance_latents=PU.rel(none, ref),
             grad_abs_max=float(raux["grad"].abs().max()), loss_value=float(raux["loss"]),
             score_over_eps=float(((1 - float(acp[int(ts[i])])) ** 0.5 * raux["grad"]).norm() / eps.norm()))
    PU.report(key, **{"guided_step9_" + k: v for k, v in e.items()})
    assert e["half_gradient_latents"] >= 1e-3, e       # guidance is visible in ONE step
    assert e["latents"] < 0.5 * e["half_gradient_latents"], e
    assert e["grad"] < 2 * PU.TOL_GRAD_CONFIG5 and e["loss"] < PU.TOL_LOSS * 2, e
    torch.cuda.empty_cache()


@pytest.mark.parametrize("variant", ["outlier_channels", "bos_token"])
def test_stress_variants_config2_shape(world, variant):
    """The stress variants at config 2's shape (16 f x 64 x 64 latent): every kernel's size-dependent path - the level-0 ring with
    four query tiles per wave, split-K convolutions, the tile-loop GEMMs - sees the outlier channels / the BOS-like token.  Forward
    (B = 2) and one guided step vs the fp32 oracle, the reference's own fp16 arithmetic as the witness of the bound
    ("sharp_attention" at this shape is test_guidance_sensitive_loop_config2)."""
    dev, cfg, sd, _, _ = world
    sd2, n = PU.stress_weights(sd, variant, dev)
    eng = UNet3DEngine(sd2, cfg, dev)
    sdo = PU.oracle_weights(sd2, dev)
    F, H, W = 16, 64, 64
    key = "cfg2_%s_stress" % variant
    lat, text, vid, noise = PU.synth_inputs(cfg, F, H, W, dev)
    if variant == "bos_token":
        text = text.clone()
        text[:, 0] *= 20.0
    smp = sampler(eng, 30, 18, 0.4)
    t0 = int(smp.timesteps[0])
    got = PU.to_lat(eng.forward(lat, t0, text, dup=True), 2, F, H, W)
    sd16 = PU.fp16_weights(sd2)
    with torch.no_grad(), PU.oracle_mode(dev):
        ref = U.unet_forward(sdo, cfg, lat.float().expand(2, -1, -1, -1, -1), t0, text.float())
        ref16 = U.unet_forward(sd16, cfg, lat.expand(2, -1, -1, -1, -1), t0, text)
    e_fwd, w_fwd = PU.rel(got, ref), PU.rel(ref16, ref)
    PU.report(key, forward_b2_rel=e_fwd, witness_fp16_oracle_forward_rel=w_fwd, eps_abs_max=ref.abs().max())
    assert torch.isfinite(got).all() and e_fwd < max(PU.TOL_FWD, 1.25 * w_fwd), (e_fwd, w_fwd)
    del got, ref, ref16
    with PU.oracle_mode(dev):
        rep_ref = G.extract_representation(sdo, cfg, vid.float(), noise.float(), text[0:1].float())
    PU.check_guided_step(eng, smp, sdo, cfg, lat, text, rep_ref, 0, key, tol_grad=PU.TOL_GRAD, witness_sd16=sd16,
                         witness_factor=1.25)
    torch.cuda.empty_cache()
