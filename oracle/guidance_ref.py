"""ORACLE - TEST INFRASTRUCTURE ONLY.  Never imported by motionclone_amd / motionclone.

CPU (PyTorch fp32 + autograd) restatement of the MotionClone guidance layer and of the DDIM
scheduler state it consumes.  The DDIM tables live in diffusers==0.16.0
(schedulers/scheduling_ddim.py, pinned at environment.yaml:13), which is NOT vendored under
/root/reference: its published algorithm is restated here and anchored on the reference's call
sites (t2v_video_sample.py:45, model_config.yaml:16-21, motionclone_functions.py:326-389).
"diffusers parity unpinned" - no copy of diffusers exists offline to check against; the reference's
own arithmetic around it is pinned by tests/test_oracle_pins.py and tests/golden/.
"""
import numpy as np
import torch
import torch.nn.functional as Fn

from . import unet3d_ref as U


# ---- scheduler state ----------------------------------------------------------------------------------
def alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    """DDIMScheduler(beta_schedule='linear') tables (model_config.yaml:16-21): cumprod(1 - linspace)."""
    betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    return torch.cumprod(1.0 - betas, dim=0)


FINAL_ALPHA_CUMPROD = 1.0  # set_alpha_to_one=True default


def uneven_timesteps(num_inference_steps, guidance_steps, guidance_scale, num_train_timesteps=1000):
    """schedule_set_timesteps(..., 'uneven') (motionclone_functions.py:432-445)."""
    split = int((1 - guidance_scale) * num_train_timesteps)
    guided = np.linspace(split, num_train_timesteps - 1, guidance_steps).round()[::-1].copy().astype(np.int64)
    plain = np.linspace(0, split - 1, num_inference_steps - guidance_steps).round()[::-1].copy().astype(np.int64)
    return np.concatenate((guided, plain))


def add_noise(acp, timestep, x0, noise):
    """add_noise (motionclone_functions.py:19-23)."""
    a = acp[timestep]
    return a ** 0.5 * x0 + (1 - a) ** 0.5 * noise


def ddim_step(acp, timesteps, step_index, eps, sample, score=None, guidance_scale=1.0):
    """schedule_customized_step with eta = 0, prediction_type 'epsilon' (motionclone_functions.py:326-389).
    x0 is formed from the un-guided eps; only the direction term sees the score."""
    t = int(timesteps[step_index])
    t_prev = int(timesteps[step_index + 1]) if step_index + 1 < len(timesteps) else -1
    a_t = acp[t]
    a_prev = acp[t_prev] if t_prev >= 0 else torch.tensor(FINAL_ALPHA_CUMPROD)
    x0 = (sample - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
    if score is not None and guidance_scale > 0.0:
        eps = eps - guidance_scale * (1 - a_t) ** 0.5 * score
    return a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps


def ddim_step_general(acp, final_acp, timesteps, step_index, model_output, sample, prediction_type="epsilon",
                      clip_sample=False, clip_sample_range=1.0, eta=0.0, use_clipped_model_output=False,
                      variance_noise=None, score=None, guidance_scale=1.0, indices=None, return_middle=False):
    """schedule_customized_step with all of its branches (motionclone_functions.py:285-409), fp32.
    Returns (prev_sample, pred_original_sample, alpha_prod_t_prev) or, for return_middle with a score (:371-372),
    (pred_epsilon, alpha_prod_t, alpha_prod_t_prev, pred_original_sample)."""
    t = int(timesteps[step_index])
    t_prev = int(timesteps[step_index + 1]) if step_index + 1 < len(timesteps) else -1      # :327
    a_t = acp[t]
    a_prev = acp[t_prev] if t_prev >= 0 else final_acp                                     # :330-331
    b_t = 1 - a_t
    if prediction_type == "epsilon":                                                       # :337-346
        x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        eps = model_output
    elif prediction_type == "sample":
        x0 = model_output
        eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
    elif prediction_type == "v_prediction":
        x0 = a_t ** 0.5 * sample - b_t ** 0.5 * model_output
        eps = a_t ** 0.5 * model_output + b_t ** 0.5 * sample
    else:
        raise ValueError(prediction_type)
    if clip_sample:                                                                        # :356-360
        x0 = x0.clamp(-clip_sample_range, clip_sample_range)
    variance = (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)                               # _get_variance, :364
    std = eta * variance ** 0.5
    if use_clipped_model_output:                                                           # :367-369
        eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
    if score is not None and return_middle:                                                # :371-372
        return eps, a_t, a_prev, x0
    if score is not None and guidance_scale > 0.0:                                         # :375-383
        if indices is not None:
            eps = eps.clone()
            eps[indices] = eps[indices] - guidance_scale * (1 - a_t) ** 0.5 * score
        else:
            eps = eps - guidance_scale * (1 - a_t) ** 0.5 * score
    prev = a_prev ** 0.5 * x0 + (1 - a_prev - std ** 2) ** 0.5 * eps                       # :386-389
    if eta > 0:                                                                            # :391-405
        prev = prev + std * variance_noise
    return prev, x0, a_prev


# ---- guidance read-out --------------------------------------------------------------------------------
def temp_attn_prob(record, heads):
    """get_temp_attn_prob (motionclone_functions.py:260-283): softmax(scale q_h k_h^T) per hooked module,
    reshaped to [(b n), heads, F, F]."""
    out = {}
    for name, (q, k) in record.items():
        BN, F, C = q.shape
        d = C // heads
        qh = q.reshape(BN, F, heads, d).transpose(1, 2)
        kh = k.reshape(BN, F, heads, d).transpose(1, 2)
        out[name] = torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, dim=-1)
    return out


def motion_representation(prob):
    """top-1 value / uint8 index per query row (motionclone_functions.py:79)."""
    rep = {}
    for name, p in prob.items():
        v, i = torch.topk(p, k=1, dim=-1)
        rep[name] = [v, i.to(torch.uint8)]
    return rep


def temp_loss(prob, rep):
    """compute_temp_loss (motionclone_functions.py:85-100): sum over modules of mse(gather(P, idx), ref)."""
    losses = []
    for name, p in prob.items():
        ref_v, ref_i = rep[name]
        cur = torch.gather(p, index=ref_i.to(torch.int64), dim=-1)
        losses.append(Fn.mse_loss(cur, ref_v.to(cur.dtype).detach()))
    return torch.stack(losses).sum()


def guidance_scale_factor(step_index, guidance_steps, warm_up_steps, cool_up_steps):
    """warm-up / cool-down multipliers (motionclone_functions.py:228-234); both may apply."""
    s = 1.0
    if step_index < warm_up_steps:
        s *= (step_index + 1) / warm_up_steps
    if step_index > guidance_steps - cool_up_steps:
        s *= (guidance_steps - step_index) / cool_up_steps
    return s


def extract_representation(sd, cfg, video_latents, noise, uncond_text, add_noise_step=400, guidance_block=1):
    """obtain_motion_representation minus VAE/CLIP (motionclone_functions.py:40-43,74-79)."""
    acp = alphas_cumprod()
    noisy = add_noise(acp, add_noise_step, video_latents, noise)
    rec = {}
    with torch.no_grad():
        U.unet_forward(sd, cfg, noisy, add_noise_step, uncond_text, guidance_block=guidance_block,
                       only_motion_feature=True, record=rec, hooked=("up_blocks.%d" % guidance_block,))
        return motion_representation(temp_attn_prob(rec, cfg["motion_heads"]))


def guided_step(sd, cfg, latents, step_index, timesteps, text, rep, hp, guidance_block=1, res_u=None, res_c=None):
    """single_step_video, guided branch (motionclone_functions.py:200-243).
    hp: dict(cfg_scale, motion_guidance_weight, guidance_steps, warm_up_steps, cool_up_steps).
    Returns (next latents, dict of intermediates for parity checks)."""
    acp = alphas_cumprod()
    t = int(timesteps[step_index])
    hooked = ("up_blocks.%d" % guidance_block,)
    control = latents.clone().detach().requires_grad_(True)
    with torch.no_grad():
        eps_u = U.unet_forward(sd, cfg, latents, t, text[[0]], guidance_block=guidance_block,
                               down_residuals=res_u[0] if res_u else None, mid_residual=res_u[1] if res_u else None)
    rec = {}
    eps_c = U.unet_forward(sd, cfg, control, t, text[[1]], guidance_block=guidance_block, record=rec, hooked=hooked,
                           down_residuals=res_c[0] if res_c else None, mid_residual=res_c[1] if res_c else None)
    prob = temp_attn_prob(rec, cfg["motion_heads"])
    loss = hp["motion_guidance_weight"] * temp_loss(prob, rep)
    loss = loss * guidance_scale_factor(step_index, hp["guidance_steps"], hp["warm_up_steps"], hp["cool_up_steps"])
    (grad,) = torch.autograd.grad(loss, control)
    eps = eps_c + hp["cfg_scale"] * (eps_c - eps_u)          # :239 (sic: eps_c + s (eps_c - eps_u))
    nxt = ddim_step(acp, timesteps, step_index, eps.detach(), control.detach(), score=grad.detach())
    return nxt.detach(), dict(eps_u=eps_u.detach(), eps_c=eps_c.detach(), loss=loss.detach(), grad=grad.detach())


def plain_step(sd, cfg, latents, step_index, timesteps, text, res=None):
    """single_step_video, un-guided branch: one B=2 UNet call on expanded latents (:245-257)."""
    acp = alphas_cumprod()
    t = int(timesteps[step_index])
    with torch.no_grad():
        eps2 = U.unet_forward(sd, cfg, latents.expand(2, -1, -1, -1, -1), t, text,
                              down_residuals=res[0] if res else None, mid_residual=res[1] if res else None)
    return eps2, acp


def plain_step_full(sd, cfg, latents, step_index, timesteps, text, cfg_scale, res=None):
    eps2, acp = plain_step(sd, cfg, latents, step_index, timesteps, text, res)
    eps = eps2[[1]] + cfg_scale * (eps2[[1]] - eps2[[0]])      # :255
    nxt = ddim_step(acp, timesteps, step_index, eps, latents, score=None)
    return nxt, dict(eps_u=eps2[[0]], eps_c=eps2[[1]])


def sample_loop(sd, cfg, latents, text, rep, hp, num_inference_steps, guidance_steps, guidance_scale,
                guidance_block=1):
    """The step loop of sample_video (motionclone_functions.py:164-166)."""
    ts = uneven_timesteps(num_inference_steps, guidance_steps, guidance_scale)
    hp = dict(hp, guidance_steps=guidance_steps)
    for i in range(len(ts)):
        if i < guidance_steps:
            latents, _ = guided_step(sd, cfg, latents, i, ts, text, rep, hp, guidance_block)
        else:
            latents, _ = plain_step_full(sd, cfg, latents, i, ts, text, hp["cfg_scale"])
    return latents
