"""Shared checks of the HIP engine against the oracle, size-agnostic: the same functions run the tiny UNet on the host
simulator (CPU suite) and the SD-1.5-width UNet at BASELINE shapes with the oracle in fp32 ON THE MI355X
(tests/test_fullsize_parity.py).  Test infrastructure only.

Inputs follow SURVEY.md 8(d): latents ~ N(0,1) seed 2025 (prepare_latents, pipeline_animation.py:316), text ~ N(0,1)
[2, n, dim] seed 7 (row 0 = uncond), reference-video latents 0.18215 N(0,1) seed 11, extraction noise seed 2025."""
import contextlib
import json
import os

import torch

from motionclone_amd import ops
from oracle import guidance_ref as G
from oracle import unet3d_ref as U

HP = dict(cfg_scale=7.5, motion_guidance_weight=2000.0, warm_up_steps=10, cool_up_steps=10)
# Relative L2, fp16 storage vs the fp32 oracle.  Round 3: set at 3-4x what is measured on MI355X (forward / latents 1.1e-3 ..
# 1.8e-3, gradient 6.4e-3 .. 7.6e-3, loss 1e-4 .. 3.3e-4, tie gap <= 3e-4, flips <= 0.31 % of the rows) so that a 3x regression
# fails; the counts move by a few rows from run to run of the ORACLE (its fp32 GEMMs pick different split-K orders per shape).
TOL_FWD, TOL_GRAD, TOL_LOSS = 5e-3, 2e-2, 2e-3
# Round 5 (verdict item 5c): the full-size cases pass their own gradient bound = 2x what is measured on MI355X - 6.5e-3 .. 7.9e-3
# at configs 1 / 2 / 4 and F = 32 -> 1.2e-2; 1.13e-2 at config 5's real size (|grad| <= 6e-6 there) -> 1.5e-2.  TOL_GRAD stays
# the bound of the tiny simulator cases (few elements: the relative L2 of a 4 x 4 map is noisier).
TOL_GRAD_FULLSIZE, TOL_GRAD_CONFIG5 = 1.2e-2, 1.5e-2
# Round 6 (ADVICE): the most an fp16 witness may loosen the gradient bound of the stress cases - 2x the worst engine error ever
# measured on them (1.64e-2, the BOS-token case)
TOL_GRAD_WITNESS_CAP = 3.3e-2
TIE_GAP = 5e-4                                      # a flipped arg-max must be a tie at this level of the fp32 oracle's P
#                                                     (round 4: 1e-3 -> 5e-4; the worst gap ever measured is 3.4e-4)
MAX_FLIP_FRACTION = 5e-3
REPORT_FILE = "parity_r06.json"

_REPORT = {}


def report(key, **vals):
    """collect measured errors; written to gpurun_out/<REPORT_FILE> (copied into profiles/ and DESIGN.md 4 by hand)"""
    _REPORT.setdefault(key, {}).update({k: (float(v) if hasattr(v, "__float__") else v) for k, v in vals.items()})
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open(os.path.join("gpurun_out", REPORT_FILE), "w") as f:
            json.dump(_REPORT, f, indent=1, sort_keys=True)
    except OSError:
        pass
    print("PARITY", key, json.dumps({k: _REPORT[key][k] for k in vals}, sort_keys=True))


@contextlib.contextmanager
def oracle_mode(dev):
    """fp32 oracle on `dev`.  On the GPU the convolutions go through PyTorch's own im2col + GEMM path (MIOpen off): the box
    is fresh, MIOpen has no precompiled gfx950 kernels and would JIT every conv shape; fp32 matmuls stay full precision."""
    if dev.type == "cuda":
        old = torch.backends.cuda.matmul.allow_tf32
        torch.backends.cuda.matmul.allow_tf32 = False
        use_miopen = os.environ.get("MC_ORACLE_MIOPEN", "0") == "1"
        with torch.backends.cudnn.flags(enabled=use_miopen, benchmark=False, deterministic=True):
            yield
        torch.backends.cuda.matmul.allow_tf32 = old
    else:
        yield


def synth_inputs(cfg, F, H, W, dev, n_text=77):
    g = lambda s: torch.Generator(device=dev).manual_seed(s)   # noqa: E731
    lat = torch.randn((1, 4, F, H, W), generator=g(2025), device=dev, dtype=torch.float32).half()
    text = torch.randn((2, n_text, cfg["cross_attention_dim"]), generator=g(7), device=dev).half()
    vid = (0.18215 * torch.randn((1, 4, F, H, W), generator=g(11), device=dev)).half()
    noise = torch.randn((1, 4, F, H, W), generator=g(2025), device=dev, dtype=torch.float32).half()
    return lat, text, vid, noise


def rel(a, b):
    a, b = a.float(), b.float().to(a.device)
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def to_lat(eps_tokens, B, F, H, W):
    return ops.cl_to_latent(eps_tokens, B, 4, F, H, W).float()


def oracle_weights(sd16, dev):
    """the engine's fp16 weights as fp32 tensors for the oracle: both sides share bit-identical parameters"""
    return {k: v.to(dev, torch.float32) for k, v in sd16.items()}


def check_forward_b2(eng, sdo, cfg, lat, text, t, key):
    B, _, F, H, W = (2,) + tuple(lat.shape[1:])
    eps = eng.forward(lat.expand(2, -1, -1, -1, -1), t, text)
    with torch.no_grad(), oracle_mode(lat.device):
        ref = U.unet_forward(sdo, cfg, lat.float().expand(2, -1, -1, -1, -1), t, text.float())
    got = to_lat(eps, 2, F, H, W)
    e = rel(got, ref)
    report(key, forward_b2_rel=e, eps_abs_max=ref.abs().max())
    assert torch.isfinite(got).all()
    assert e < TOL_FWD, e
    return got, ref


def flip_stats(idx, val, prob):
    """idx/val: engine top-1 [BN, heads, F, 1]; prob: oracle P fp32 [BN, heads, F, F].
    -> (#flips, total, max oracle gap over flips, max |val - oracle top value|)"""
    top_v, top_i = torch.topk(prob, k=1, dim=-1)
    idx = idx.to(prob.device).long()
    mism = idx != top_i
    alt = torch.gather(prob, -1, idx)
    gap = (top_v - alt)[mism]
    return int(mism.sum()), mism.numel(), (float(gap.max()) if gap.numel() else 0.0), \
        float((val.float().to(prob.device) - top_v).abs().max())


def check_extraction(eng, smp, sdo, cfg, vid, noise, text, key, ctrl=None, res=None):
    """obtain_motion_representation's model part: exact flip count of the uint8 arg-max, every flip proven a tie"""
    rep = smp.extract(vid, noise, text[0:1], add_noise_step=400, ctrl=ctrl)
    noisy = smp.add_noise(400, vid, noise).float()
    rec = {}
    with torch.no_grad(), oracle_mode(vid.device):
        U.unet_forward(sdo, cfg, noisy, 400, text[0:1].float(), only_motion_feature=True, record=rec,
                       down_residuals=res[0] if res else None, mid_residual=res[1] if res else None)
        prob = G.temp_attn_prob(rec, cfg["motion_heads"])
    ref = G.motion_representation(prob)
    assert list(rep) == list(ref)
    flips = total = 0
    worst_gap = worst_val = 0.0
    for k in ref:
        v, i = rep[k]
        assert v.shape == ref[k][0].shape and i.dtype == torch.uint8
        n, tot, gap, dv = flip_stats(i, v, prob[k])
        flips, total = flips + n, total + tot
        worst_gap, worst_val = max(worst_gap, gap), max(worst_val, dv)
    report(key, extraction_flips=flips, extraction_rows=total, extraction_flip_max_gap=worst_gap,
           extraction_value_abs_max_err=worst_val)
    assert worst_gap <= TIE_GAP, "an arg-max flip that is not a tie: gap %g" % worst_gap
    assert worst_val < 5e-3
    assert flips <= MAX_FLIP_FRACTION * total, (flips, total)
    return rep, ref, prob


def fp16_weights(sd):
    """the reference's own arithmetic: the same parameters in fp16 for the oracle run in fp16 through stock PyTorch-ROCm"""
    return {k: v.half() for k, v in sd.items()}


def check_guided_step(eng, smp, sdo, cfg, lat, text, rep_ref, step_index, key, ctrl=None, res_u=None, res_c=None,
                      tol_grad=None, witness_sd16=None, witness_factor=None):
    """witness_sd16 (round 5, SURVEY.md 8c "second witness"): the oracle once more in fp16 on the device (= what the reference
    computes through stock PyTorch: fp16 activations AND an fp16 autograd backward) - its loss / gradient / latents are
    reported next to the engine's, both measured from the fp32 oracle.
    witness_factor (stress cases on ill-conditioned weights): every bound becomes max(the usual tolerance, factor x the fp16
    reference's own distance from the fp32 oracle) - "within fp16 tolerance" measured on the spot; the gradient against HALF the
    witness's (fp16 autograd is the worse of the two by an order of magnitude)."""
    tol_grad = TOL_GRAD if tol_grad is None else tol_grad
    ts = G.uneven_timesteps(smp.N, smp.G, smp_guidance_scale(smp))
    assert smp.timesteps.tolist() == ts.tolist()
    hp = dict(HP, guidance_steps=smp.G)
    B, _, F, H, W = lat.shape
    aux = {}
    nxt = smp.step(lat, step_index, text, eng.prepare_representation(rep_ref), aux=aux, ctrl=ctrl)
    with oracle_mode(lat.device):
        ref_nxt, ref_aux = G.guided_step(sdo, cfg, lat.float(), step_index, ts, text.float(), rep_ref, hp,
                                         res_u=res_u, res_c=res_c)
    e = dict(eps_c=rel(to_lat(aux["eps_c"], 1, F, H, W), ref_aux["eps_c"]),
             eps_u=rel(to_lat(aux["eps_u"], 1, F, H, W), ref_aux["eps_u"]),
             loss=abs(float(aux["loss"]) - float(ref_aux["loss"])) / abs(float(ref_aux["loss"])),
             grad=rel(aux["grad"], ref_aux["grad"]), latents=rel(nxt, ref_nxt),
             loss_value=float(ref_aux["loss"]), grad_abs_max=float(ref_aux["grad"].abs().max()))
    report(key, **{"guided_" + k: v for k, v in e.items()})
    if witness_sd16 is not None:
        h = lambda r: None if r is None else ([t.half() for t in r[0]], r[1].half())   # noqa: E731
        rep16 = {k: [v[0].half(), v[1]] for k, v in rep_ref.items()}
        with oracle_mode(lat.device):
            w_nxt, w_aux = G.guided_step(witness_sd16, cfg, lat.half(), step_index, ts, text.half(), rep16, hp,
                                         res_u=h(res_u), res_c=h(res_c))
        gnorm = float(w_aux["grad"].float().norm())
        wit = dict(loss=abs(float(w_aux["loss"]) - float(ref_aux["loss"])) / abs(float(ref_aux["loss"])),
                   grad=rel(w_aux["grad"], ref_aux["grad"]) if gnorm == gnorm else float("nan"),
                   latents=rel(w_nxt, ref_nxt), eps_c=rel(w_aux["eps_c"], ref_aux["eps_c"]),
                   grad_finite=bool(torch.isfinite(w_aux["grad"]).all()),
                   grad_zero_fraction=float((w_aux["grad"] == 0).float().mean()))
        report(key, **{"witness_fp16_oracle_guided_" + k: v for k, v in wit.items()})
        del w_nxt, w_aux
    assert torch.isfinite(aux["grad"]).all() and torch.isfinite(nxt.float()).all()
    t_eps = t_lat = TOL_FWD
    t_loss = TOL_LOSS
    if witness_factor is not None:
        assert witness_sd16 is not None
        t_eps = max(TOL_FWD, witness_factor * wit["eps_c"])
        t_lat = max(TOL_FWD, witness_factor * wit["latents"])
        t_loss = max(TOL_LOSS, witness_factor * wit["loss"])
        # the fp16 witness's gradient is 0.10-0.14 off fp32 on the ill-conditioned cases (underflow): half of THAT would admit a 4x
        # regression of the engine (measured 0.006-0.016).  The witness may loosen the bound only up to TOL_GRAD_WITNESS_CAP.
        if wit["grad"] == wit["grad"]:
            tol_grad = max(tol_grad, min(0.5 * wit["grad"], TOL_GRAD_WITNESS_CAP))
    assert e["eps_c"] < t_eps and e["eps_u"] < t_eps, e
    assert e["loss"] < t_loss, e
    assert e["grad"] < tol_grad, e
    assert e["latents"] < t_lat, e
    return nxt, ref_nxt


def smp_guidance_scale(smp):
    return smp.guidance_scale


def check_plain_step(eng, smp, sdo, cfg, lat, text, step_index, key, ctrl=None, res=None):
    ts = G.uneven_timesteps(smp.N, smp.G, smp_guidance_scale(smp))
    nxt = smp.step(lat, step_index, text, {}, ctrl=ctrl)
    with oracle_mode(lat.device):
        ref_nxt, _ = G.plain_step_full(sdo, cfg, lat.float(), step_index, ts, text.float(), HP["cfg_scale"], res=res)
    e = rel(nxt, ref_nxt)
    report(key, **{"plain_step_%d_latents" % step_index: e})
    assert e < TOL_FWD, e
    return nxt, ref_nxt


def check_loop(eng, smp, sdo, cfg, lat, text, rep_ref, key, tol, first=0, last=None, start_ref=None, ctrl=None, csdo=None,
               witness_sd16=None):
    """Steps [first, last) of the loop (sample_video, motionclone_functions.py:164-166): engine and oracle each follow
    their OWN trajectory from a common start (`start_ref`: fp32 latents at step `first`, default the initial noise).
    ctrl / csdo: SparseCtrl condition and the ControlNet's oracle weights - the encoder runs every step on the current
    timestep (motionclone_functions.py:176-197), on both sides.
    witness_sd16 (round 5): a THIRD trajectory - the oracle in fp16 (the reference's arithmetic through stock PyTorch), its
    drift from the fp32 oracle reported per step next to the engine's (not with SparseCtrl)."""
    ts = G.uneven_timesteps(smp.N, smp.G, smp_guidance_scale(smp))
    hp = dict(HP, guidance_steps=smp.G)
    rep_dev = eng.prepare_representation(rep_ref)
    last = smp.N if last is None else last
    xr = lat.float() if start_ref is None else start_ref.float()
    x = xr.half()
    drift = []
    wdrift = []
    xw = xr.half() if witness_sd16 is not None else None
    rep16 = {k: [v[0].half(), v[1]] for k, v in rep_ref.items()} if witness_sd16 is not None else None
    shape2 = (2,) + tuple(lat.shape[1:])
    for i in range(first, last):
        x = smp.step(x, i, text, rep_dev, ctrl=ctrl)
        with oracle_mode(lat.device):
            d = m = None
            if ctrl is not None:
                with torch.no_grad():
                    d, m = U.controlnet_forward(csdo, cfg, shape2, int(ts[i]), text.float(), ctrl["cond"].float(),
                                                ctrl["mask"].float(), ctrl.get("scale", 1.0))
            if i < smp.G:
                xr, _ = G.guided_step(sdo, cfg, xr, i, ts, text.float(), rep_ref, hp,
                                      res_u=([t[[0]] for t in d], m[[0]]) if d else None,
                                      res_c=([t[[1]] for t in d], m[[1]]) if d else None)
            else:
                xr, _ = G.plain_step_full(sdo, cfg, xr, i, ts, text.float(), hp["cfg_scale"], res=(d, m) if d else None)
            if xw is not None:
                assert ctrl is None
                if i < smp.G:
                    xw, _ = G.guided_step(witness_sd16, cfg, xw, i, ts, text.half(), rep16, hp)
                else:
                    xw, _ = G.plain_step_full(witness_sd16, cfg, xw, i, ts, text.half(), hp["cfg_scale"])
                wdrift.append(rel(xw, xr))
        drift.append(rel(x, xr))
    report(key, loop_drift=[round(d, 6) for d in drift], loop_steps=[first, last], loop_schedule=[smp.N, smp.G])
    if wdrift:
        report(key, witness_fp16_oracle_loop_drift=[round(d, 6) for d in wdrift],
               witness_fp16_oracle_finite=bool(torch.isfinite(xw.float()).all()))
    assert torch.isfinite(x.float()).all()
    assert max(drift) < tol, drift
    return drift


# ---- round 6: guidance that MATTERS ----------------------------------------------------------------------------------------
def stress_weights(sd, variant, dev, seed=99):
    """The stress variants of tests/test_fullsize_parity.py as a weight transform (seeded; returns the new state dict and the
    counts of what was touched): "outlier_channels" - 4 channels of every GroupNorm / LayerNorm gain x 12 and two output channels
    of every FeedForward x 6; "sharp_attention" - every to_q / to_k x 2.5 (logits span 6 x what N(0, init) gives: the temporal maps
    turn nearly one-hot and the MotionClone loss / gradient become large); "bos_token" - weights unchanged (the text is)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    sd2 = {}
    n = dict(norm=0, ff=0, qk=0)
    for k, v in sd.items():
        v = v.clone()
        if variant == "outlier_channels":
            if k.endswith("weight") and v.dim() == 1 and ("norm" in k):
                idx = torch.randperm(v.numel(), generator=g, device=dev)[:4]
                v[idx] = v[idx] * 12.0
                n["norm"] += 1
            elif k.endswith("ff.net.2.weight"):
                idx = torch.randperm(v.shape[0], generator=g, device=dev)[:2]
                v[idx] = v[idx] * 6.0
                n["ff"] += 1
        elif variant == "sharp_attention" and (k.endswith("to_q.weight") or k.endswith("to_k.weight")):
            v = v * 2.5
            n["qk"] += 1
        sd2[k] = v
    return sd2, n


def oracle_guided_step_scaled(sdo, cfg, x, i, ts, text, rep, hp, grad_factor):
    """The oracle's guided step with the MotionClone gradient multiplied by `grad_factor` before the DDIM update (1 = the
    reference; 0.5 = what a factor-2 error anywhere in the guidance gradient path would do; 0 = guidance dropped)."""
    nxt, aux = G.guided_step(sdo, cfg, x, i, ts, text, rep, hp)
    if grad_factor != 1.0:
        eps = aux["eps_c"] + hp["cfg_scale"] * (aux["eps_c"] - aux["eps_u"])
        nxt = G.ddim_step(G.alphas_cumprod(), ts, i, eps, x, score=grad_factor * aux["grad"])
    return nxt, aux


def check_guidance_sensitive_loop(eng, smp, sdo, cfg, lat, text, rep_ref, key, last=None, margin=0.5, min_visibility=3e-3):
    """Steps [0, last) with FOUR trajectories from the same start: the engine, the fp32 oracle, the oracle with the guidance
    gradient HALVED (a factor-2 error of the gradient path) and the oracle with guidance DROPPED.  The engine must stay closer to
    the oracle than `margin` x the halved-gradient trajectory does - at every guided step from the third on and at the end - so a
    gradient path that is off by 2x FAILS; and the halved trajectory must be at least `min_visibility` away (otherwise the weights
    make guidance invisible and the test proves nothing: that is an error of the test, not a pass)."""
    ts = G.uneven_timesteps(smp.N, smp.G, smp_guidance_scale(smp))
    hp = dict(HP, guidance_steps=smp.G)
    rep_dev = eng.prepare_representation(rep_ref)
    last = smp.N if last is None else last
    xr = lat.float()
    xh, x0g, x = xr.clone(), xr.clone(), xr.half()
    rows = []
    for i in range(last):
        x = smp.step(x, i, text, rep_dev)
        with oracle_mode(lat.device):
            if i < smp.G:
                xr, aux = oracle_guided_step_scaled(sdo, cfg, xr, i, ts, text.float(), rep_ref, hp, 1.0)
                xh, _ = oracle_guided_step_scaled(sdo, cfg, xh, i, ts, text.float(), rep_ref, hp, 0.5)
                x0g, _ = oracle_guided_step_scaled(sdo, cfg, x0g, i, ts, text.float(), rep_ref, hp, 0.0)
                eps = aux["eps_c"] + hp["cfg_scale"] * (aux["eps_c"] - aux["eps_u"])
                a_t = float(G.alphas_cumprod()[int(ts[i])])
                score_over_eps = float(((1 - a_t) ** 0.5 * aux["grad"]).norm() / eps.norm())
            else:
                xr, _ = G.plain_step_full(sdo, cfg, xr, i, ts, text.float(), hp["cfg_scale"])
                xh, _ = G.plain_step_full(sdo, cfg, xh, i, ts, text.float(), hp["cfg_scale"])
                x0g, _ = G.plain_step_full(sdo, cfg, x0g, i, ts, text.float(), hp["cfg_scale"])
                score_over_eps = None
        rows.append(dict(step=i, engine=rel(x, xr), half_gradient=rel(xh, xr), no_guidance=rel(x0g, xr),
                         score_over_eps=score_over_eps))
    report(key, sensitive_loop=[{k: (round(v, 6) if isinstance(v, float) else v) for k, v in r.items()} for r in rows],
           loop_schedule=[smp.N, smp.G], loop_steps=[0, last])
    assert torch.isfinite(x.float()).all()
    vis = max(r["half_gradient"] for r in rows)
    assert vis >= min_visibility, "guidance is numerically invisible with these weights (half-gradient trajectory %.2e away)" % vis
    for r in rows:
        if (2 <= r["step"] < smp.G) or r["step"] == last - 1:
            assert r["engine"] < margin * r["half_gradient"], r
    return rows
