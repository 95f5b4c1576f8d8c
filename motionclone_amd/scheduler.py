"""Minimal DDIM scheduler state for boxes without diffusers: exactly the attributes the path reads from
`diffusers.DDIMScheduler`: alphas_cumprod, final_alpha_cumprod, init_noise_sigma, config,
num_inference_steps, timesteps, scale_model_input, step signature (eta, generator)."""
import types

import torch

from .sampler import ddim_alphas_cumprod


class DDIMSchedulerState:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear",
                 steps_offset=1, clip_sample=False, set_alpha_to_one=True, prediction_type="epsilon",
                 clip_sample_range=1.0, thresholding=False, **unused):
        if beta_schedule != "linear":
            raise NotImplementedError("the path is configured with linear betas (configs/model_config/model_config.yaml:16-21)")
        if prediction_type not in ("epsilon", "sample", "v_prediction"):
            raise ValueError(f"prediction_type given as {prediction_type} must be one of `epsilon`, `sample`, or `v_prediction`")
        if thresholding:
            raise NotImplementedError("dynamic thresholding (per-sample quantiles) is not on the MotionClone path")
        self.alphas_cumprod = ddim_alphas_cumprod(num_train_timesteps, beta_start, beta_end)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = None
        self.config = types.SimpleNamespace(num_train_timesteps=num_train_timesteps, steps_offset=steps_offset,
                                            clip_sample=clip_sample, clip_sample_range=clip_sample_range,
                                            prediction_type=prediction_type, thresholding=False)
        self.variance_type = "fixed_small"

    def _get_variance(self, timestep, prev_timestep):
        """diffusers 0.16.0 DDIMScheduler._get_variance: sigma_t^2 of DDIM formula (16) at eta = 1"""
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        return (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, eta=0.0, generator=None):
        raise NotImplementedError("use customized_step (schedule_customized_step) as the reference does")
