"""Pipeline shell (reference motionclone/pipelines/pipeline_animation.py:46-324): holds the components and the
host-side helpers around the step loop.  Stays Python; VAE / CLIP are whatever objects the caller passes (diffusers /
transformers modules when available).  The reference's own `__call__` is dead code and is not
provided."""
import contextlib
import inspect

import torch


class AnimationPipeline:
    def __init__(self, vae=None, text_encoder=None, tokenizer=None, unet=None, scheduler=None, controlnet=None):
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer
        self.unet, self.scheduler, self.controlnet = unet, scheduler, controlnet
        vae_cfg = getattr(vae, "config", None)
        n_down = len(getattr(vae_cfg, "block_out_channels", (0, 0, 0, 0))) if vae_cfg is not None else 4
        self.vae_scale_factor = 2 ** (n_down - 1)
        self.input_config = None

    def to(self, device=None, dtype=None):
        for m in (self.vae, self.text_encoder, self.unet, self.controlnet):
            if m is not None and hasattr(m, "to"):
                m.to(device) if dtype is None else m.to(device=device, dtype=dtype)
        return self

    @property
    def device(self):
        return self.unet.device

    @property
    def _execution_device(self):
        return self.unet.device

    @contextlib.contextmanager
    def progress_bar(self, total=None):
        try:
            from tqdm import tqdm
            bar = tqdm(total=total)
        except Exception:  # pragma: no cover
            bar = None

        class _B:
            def update(self_inner, n=1):
                if bar is not None:
                    bar.update(n)
        try:
            yield _B()
        finally:
            if bar is not None:
                bar.close()

    def _encode_prompt(self, prompt, device, num_videos_per_prompt, do_classifier_free_guidance, negative_prompt):
        """reference :160-247 -> [uncond, cond] embeddings [2, 77, dim]"""
        def enc(texts):
            tok = self.tokenizer(texts, padding="max_length", max_length=self.tokenizer.model_max_length,
                                 truncation=True, return_tensors="pt")
            return self.text_encoder(tok.input_ids.to(device))[0]
        prompts = [prompt] if isinstance(prompt, str) else list(prompt)
        text_embeddings = enc(prompts).repeat_interleave(num_videos_per_prompt, dim=0)
        if do_classifier_free_guidance:
            if negative_prompt is None:
                neg = [""] * len(prompts)
            elif isinstance(negative_prompt, str):
                neg = [negative_prompt] * len(prompts)
            else:
                neg = list(negative_prompt)
            if len(neg) != len(prompts):
                raise ValueError(f"`negative_prompt`: {negative_prompt} has batch size {len(neg)}, but `prompt`: "
                                 f"{prompt} has batch size {len(prompts)}.")
            uncond = enc(neg).repeat_interleave(num_videos_per_prompt, dim=0)
            text_embeddings = torch.cat([uncond, text_embeddings])
        return text_embeddings

    @torch.no_grad()
    def decode_latents(self, latents):
        """reference :249-263: frame-wise VAE decode -> float32 numpy [B, 3, F, H, W] in [0, 1]"""
        video_length = latents.shape[2]
        if hasattr(self.vae, "decode_video") and latents.shape[0] == 1:
            # native AutoencoderKL: frames batched, 1/0.18215 and the (x/2+0.5).clamp(0,1) tail fused into the HIP path
            return self.vae.decode_video(latents).cpu().float().numpy()
        latents = 1 / 0.18215 * latents
        B = latents.shape[0]
        frames = latents.permute(0, 2, 1, 3, 4).reshape(B * video_length, *latents.shape[1:2], *latents.shape[3:])
        video = torch.cat([self.vae.decode(frames[i:i + 1]).sample for i in range(frames.shape[0])])
        video = video.reshape(B, video_length, *video.shape[1:]).permute(0, 2, 1, 3, 4)
        video = (video / 2 + 0.5).clamp(0, 1)
        return video.cpu().float().numpy()

    def prepare_extra_step_kwargs(self, generator, eta):
        """reference :265-280"""
        kw = {}
        params = set(inspect.signature(self.scheduler.step).parameters.keys())
        if "eta" in params:
            kw["eta"] = eta
        if "generator" in params:
            kw["generator"] = generator
        return kw

    def prepare_latents(self, batch_size, num_channels_latents, video_length, height, width, dtype, device, generator,
                        latents=None):
        """reference :297-324"""
        shape = (batch_size, num_channels_latents, video_length, height // self.vae_scale_factor,
                 width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an "
                             f"effective batch size of {batch_size}. Make sure the batch size matches the length of "
                             f"the generators.")
        device = torch.device(device)
        if latents is None:
            if isinstance(generator, list):
                latents = torch.cat([torch.randn(shape, generator=g, device=device, dtype=dtype) for g in generator], 0)
            else:
                latents = torch.randn(shape, generator=generator, device=device, dtype=dtype)
        else:
            if latents.shape != shape:
                raise ValueError(f"Unexpected latents shape, got {latents.shape}, expected {shape}")
            latents = latents.to(device)
        return latents * self.scheduler.init_noise_sigma
