"""Guided DDIM sampler over `UNet3DEngine`: the loop of sample_video / single_step_video
(reference motionclone/utils/motionclone_functions.py:102-257) with the scheduler arithmetic of
schedule_set_timesteps (:413-472) and schedule_customized_step (:285-409) folded into one fused
CFG + DDIM kernel per step.  Host side is pure orchestration: per step it computes six scalars.
"""
import threading

import numpy as np
import torch

from . import ops

# One hipGraph capture at a time per process: HIP refuses a capture_begin while another thread's capture is open
# (hipErrorIllegalState on ROCm 7.2, also in thread-local capture mode).  The launcher additionally runs every lane's first
# example - where all captures happen - alone (motionclone_amd/lanes.py).
_CAPTURE_LOCK = threading.RLock()


def ddim_alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    """DDIMScheduler(beta_schedule='linear') state as configured at configs/model_config/model_config.yaml:16-21
    (diffusers 0.16.0 semantics: fp32 linspace -> cumprod); final_alpha_cumprod = 1 (set_alpha_to_one)."""
    betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    return torch.cumprod(1.0 - betas, dim=0)


def uneven_timesteps(num_inference_steps, guidance_steps, guidance_scale, num_train_timesteps=1000):
    """timestep_spacing_type == 'uneven' (motionclone_functions.py:432-445): dense steps while guiding."""
    if num_inference_steps > num_train_timesteps:
        raise ValueError("num_inference_steps %d exceeds num_train_timesteps %d" % (num_inference_steps,
                                                                                  num_train_timesteps))
    split = int((1 - guidance_scale) * num_train_timesteps)
    a = np.linspace(split, num_train_timesteps - 1, guidance_steps).round()[::-1].copy().astype(np.int64)
    b = np.linspace(0, split - 1, num_inference_steps - guidance_steps).round()[::-1].copy().astype(np.int64)
    return np.concatenate((a, b))


class MotionCloneSampler:
    def __init__(self, engine, cfg_scale=7.5, motion_guidance_weight=2000.0, warm_up_steps=10, cool_up_steps=10,
                 num_inference_steps=30, guidance_steps=18, guidance_scale=0.4, score_guidance_scale=1.0,
                 controlnet=None, batch_guided=True, timesteps=None, alphas_cumprod=None, final_alpha_cumprod=1.0):
        """`timesteps` / `alphas_cumprod` / `final_alpha_cumprod`: the state of an already configured scheduler (what the
        reference reads at motionclone_functions.py:213-214,326-336); by default the 'uneven' table of the arguments and the
        DDIM tables of configs/model_config/model_config.yaml:16-21."""
        self.engine = engine
        self.dev = engine.dev
        # True: guided steps run eps_u / eps_c as one B = 2 forward (same per-sample arithmetic as the reference's
        # two B = 1 calls, fewer and larger launches); False: two separate forwards exactly as the reference issues them
        self.batch_guided = batch_guided
        self.controlnet = controlnet     # ControlNetEngine or None (image-to-video, SparseCtrl)
        self.cfg_scale = float(cfg_scale)
        self.weight = float(motion_guidance_weight)
        self.warm, self.cool = warm_up_steps, cool_up_steps
        self.N, self.G = num_inference_steps, guidance_steps
        self.guidance_scale = float(guidance_scale)   # fraction of the train timesteps covered by the guided steps
        if timesteps is None:
            self.timesteps = uneven_timesteps(num_inference_steps, guidance_steps, guidance_scale)
        else:
            self.timesteps = np.asarray(torch.as_tensor(timesteps).cpu(), dtype=np.int64)
            self.N = len(self.timesteps)
        self.acp = ddim_alphas_cumprod() if alphas_cumprod is None else torch.as_tensor(alphas_cumprod).detach().float().cpu()
        self.final_alpha = float(final_alpha_cumprod)
        self.score_gs = float(score_guidance_scale)
        self._graphs = None      # step index -> (hipGraph, static input, static output); see enable_graphs()
        self._graph_pool = None
        self._warm_kinds = set()  # {(guided?, SparseCtrl?, latent shape, text shape, GEMM share)} kinds of step that already ran once eagerly on this sampler

    def _gemm_share(self):
        """the GEMM tile policy this sampler's launches run under (its / its engine's `gemm_lanes`, else the process default)"""
        lanes = getattr(self, "gemm_lanes", None)
        if lanes is None:
            lanes = getattr(self.engine, "gemm_lanes", None)
        return ops.gemm_share() if lanes is None else ops._share_of(lanes)

    def _alphas(self, i):
        t = int(self.timesteps[i])
        t_prev = int(self.timesteps[i + 1]) if i + 1 < len(self.timesteps) else -1
        a_t = float(self.acp[t])
        a_prev = float(self.acp[t_prev]) if t_prev >= 0 else self.final_alpha
        return t, a_t, a_prev

    def guidance_factor(self, i):
        """warm-up / cool-down of the guidance loss (motionclone_functions.py:228-234; strict inequalities)"""
        s = 1.0
        if i < self.warm:
            s *= (i + 1) / self.warm
        if i > self.G - self.cool:
            s *= (self.G - i) / self.cool
        return s

    def add_noise(self, t, x0, noise):
        """add_noise (motionclone_functions.py:19-23)"""
        a = float(self.acp[int(t)])
        return (a ** 0.5 * x0.float() + (1 - a) ** 0.5 * noise.float()).to(x0.dtype)

    @ops.scoped
    def extract(self, video_latents, noise, uncond_text, add_noise_step=400, ctrl=None):
        """ctrl = dict(cond, mask, scale) runs the SparseCtrl encoder first (motionclone_functions.py:46-72)"""
        noisy = self.add_noise(add_noise_step, video_latents, noise)
        if ctrl is None:
            return self.engine.extract_representation(noisy, add_noise_step, uncond_text)
        down, mid = self.controlnet.forward(tuple(noisy.shape), add_noise_step, uncond_text, ctrl["cond"], ctrl["mask"],
                                            ctrl.get("scale", 1.0))
        return self.engine.extract_representation(noisy, add_noise_step, uncond_text, down_residuals=down,
                                                  mid_residual=mid)

    # ---- hipGraph replay of whole steps (step-loop host overhead) ----------------------------
    def enable_graphs(self):
        """Capture every DDIM step (one B = 2 forward [+ guidance backward] + the fused update = ~800-1500 launches) into
        a hipGraph the first time it runs with a given (text, representation, control) and replay it afterwards: the
        launch sequence of a step is fixed, only the latent changes, and it enters through a static buffer.  All graphs
        share one memory pool (they never run concurrently).  Results are bit-identical to the eager path."""
        self._graphs = {}
        return self

    def _graphed_step(self, latents, i, text, rep_dev, ctrl):
        """One hipGraph per (step index, shapes).  Everything that changes between videos - latents, text embeddings, the
        motion representation, the SparseCtrl condition - enters through static buffers that are refreshed by copies before
        the replay, so a graph captured on the first video serves every later (prompt, reference-video) pair."""
        guided = i < self.G
        rsig = tuple((k, tuple(v[0].shape)) for k, v in rep_dev.items()) if guided else ()
        csig = None if ctrl is None else (tuple(ctrl["cond"].shape), float(ctrl.get("scale", 1.0)))
        key = (i, tuple(latents.shape), tuple(text.shape), rsig, csig, self._gemm_share())   # the GEMM geometry is baked in
        ent = self._graphs.get(key)
        if ent is None:
            from . import lanes
            if not lanes.may_capture():
                return None      # lanes already run concurrently: no capture now (lanes.may_capture), the caller goes eager
            with _CAPTURE_LOCK:
                s_lat, s_text = latents.clone(), text.clone()
                s_rep = {k: tuple(t.clone() for t in v) for k, v in rep_dev.items()} if guided else {}
                s_ctrl = None if ctrl is None else dict(cond=ctrl["cond"].clone(), mask=ctrl["mask"].clone(), scale=ctrl.get("scale", 1.0))
                # an eager pass before the FIRST capture of each kind of step (guided / plain): lazy one-time work (packed
                # weights, function attributes, workspace caches) must not happen inside a capture.  Later captures of the same
                # kind skip it - an eager guided step allocates its whole tape from the ordinary caching pool, and doing that
                # for all 30 step indices of every lane is what held 63 GiB reserved for 18 GiB in use (round 3).
                # the graph key without the step index: a new resolution, batch size or GEMM share setting touches kernels and
                # GEMM geometries (e.g. the two-workgroup tiles, chosen only for share 0) that have not run eagerly yet
                kind = (guided, ctrl is not None, tuple(latents.shape), tuple(text.shape), self._gemm_share())
                if kind not in self._warm_kinds:
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        self._step_eager(s_lat, i, s_text, s_rep, None, s_ctrl)
                    torch.cuda.current_stream().wait_stream(side)
                    self._warm_kinds.add(kind)
                graph = torch.cuda.CUDAGraph()
                # thread_local: other host threads (launcher lanes) keep allocating / launching while this one captures
                with torch.cuda.graph(graph, pool=self._graph_pool, capture_error_mode="thread_local"):
                    s_out = self._step_eager(s_lat, i, s_text, s_rep, None, s_ctrl)
                if self._graph_pool is None:
                    self._graph_pool = graph.pool()
            self._graphs[key] = (graph, s_lat, s_text, s_rep, s_ctrl, s_out)
            graph.replay()       # the step's result comes from the graph it will be replayed from (bit-identical to eager)
            return s_out
        graph, s_lat, s_text, s_rep, s_ctrl, s_out = ent
        s_lat.copy_(latents)
        if text.data_ptr() != s_text.data_ptr():
            s_text.copy_(text)
        for k, v in s_rep.items():
            for dst, src in zip(v, rep_dev[k]):
                if src.data_ptr() != dst.data_ptr():
                    dst.copy_(src)
        if s_ctrl is not None:
            for k in ("cond", "mask"):
                if ctrl[k].data_ptr() != s_ctrl[k].data_ptr():
                    s_ctrl[k].copy_(ctrl[k])
        graph.replay()
        return s_out

    def step(self, latents, i, text, rep_dev, aux=None, ctrl=None, eta=0.0, generator=None, variance_noise=None):
        """single_step_video (motionclone_functions.py:173-257): text = [uncond, cond] embeddings [2, n, dim];
        ctrl = dict(cond, mask, scale) enables the SparseCtrl pass of :176-197 (one B=2 encoder run per step);
        eta / generator / variance_noise = the `extra_step_kwargs` handed on to schedule_customized_step (:241,255):
        eta > 0 adds eta * sigma_t * noise, drawn like randn_tensor unless given (never captured in a graph).
        ALIASING: with `enable_graphs()` a replayed step returns the graph's static output buffer - valid until the same
        step index is replayed again (the next video on this sampler); callers that keep a result across videos clone it
        (the drop-in `single_step_video` / `sample_video` do)."""
        if eta:
            if variance_noise is not None and generator is not None:
                raise ValueError("Cannot pass both generator and variance_noise. Please make sure that either `generator` or"
                                 " `variance_noise` stays `None`.")
            t, a_t, a_prev = self._alphas(i)
            sigma = float(eta) * max((1.0 - a_prev) / (1.0 - a_t) * (1.0 - a_t / a_prev), 0.0) ** 0.5
            prev = self._step_eager(latents, i, text, rep_dev, aux, ctrl, sigma)
            if variance_noise is None:
                gdev = generator.device if generator is not None else latents.device
                variance_noise = torch.randn(latents.shape, generator=generator, device=gdev, dtype=latents.dtype).to(latents.device)
            flat = prev.reshape(-1, 8)
            ops.add(flat, variance_noise.to(prev.dtype).contiguous().reshape(flat.shape), out=flat, sa=1.0, sb=sigma)
            return prev
        if self._graphs is not None and aux is None and latents.is_cuda:
            out = self._graphed_step(latents, i, text, rep_dev, ctrl)
            if out is not None:
                return out
        return self._step_eager(latents, i, text, rep_dev, aux, ctrl)

    @ops.scoped
    def _step_eager(self, latents, i, text, rep_dev, aux=None, ctrl=None, sigma=0.0):
        """latents [V, 4, F, H, W]; text [2 V, n, dim] ordered [u_1 .. u_V | c_1 .. c_V] (V = 1: [uncond, cond], the reference's
        layout); rep_dev: engine.prepare_representation of the V representations (a list for V > 1).  V > 1 = V independent
        videos through ONE launch sequence (same kernels, V times the rows): each video's arithmetic is its own - no
        reduction crosses the batch - but the GEMM tile / split-K choice follows the larger row count, so results agree
        with the one-video path to fp16 rounding, not bit for bit."""
        from .engine import split_residuals
        eng = self.engine
        V = latents.shape[0]
        if text.shape[0] != 2 * V:
            raise ValueError("text must hold [uncond x V | cond x V] embeddings: %s for %d videos" % (tuple(text.shape), V))
        if V > 1 and (ctrl is not None or not self.batch_guided):
            raise NotImplementedError("several videos per step: not with SparseCtrl / batch_guided=False")
        t, a_t, a_prev = self._alphas(i)
        down = mid = None
        if ctrl is not None:
            shape2 = (2,) + tuple(latents.shape[1:])
            down, mid = self.controlnet.forward(shape2, t, text, ctrl["cond"], ctrl["mask"], ctrl.get("scale", 1.0))

        def update(eps_c, eps_u, grad, coef):
            if V == 1:
                return ops.cfg_ddim_step(eps_c, eps_u, latents, grad, self.cfg_scale, a_t, a_prev, coef, sigma=sigma)
            T1 = eps_c.shape[0] // V           # the fused CFG + DDIM update works on one video's rows at a time
            return torch.cat([ops.cfg_ddim_step(eps_c[v * T1:(v + 1) * T1], eps_u[v * T1:(v + 1) * T1], latents[v:v + 1],
                                                None if grad is None else grad[v:v + 1], self.cfg_scale, a_t, a_prev, coef,
                                                sigma=sigma) for v in range(V)], 0)
        if i < self.G:
            w = self.weight * self.guidance_factor(i)
            if self.batch_guided:
                # eps_u and eps_c from ONE B = 2 V forward; only the conditional halves are differentiated
                eps_c, grad, loss, eps_u = eng.guided_eps_and_grad(latents, t, text[V:], rep_dev, w,
                                                                   want_loss=aux is not None, down_residuals=down,
                                                                   mid_residual=mid, text_uncond=text[:V])
            else:
                du = mu = dc = mc = None
                if down is not None:
                    du, mu = split_residuals(down, mid, 0, 2)
                    dc, mc = split_residuals(down, mid, 1, 2)
                eps_u = eng.forward(latents, t, text[0:1], down_residuals=du, mid_residual=mu)
                eps_c, grad, loss = eng.guided_eps_and_grad(latents, t, text[1:2], rep_dev, w, want_loss=aux is not None,
                                                            down_residuals=dc, mid_residual=mc)
            if aux is not None:
                aux.update(eps_u=eps_u, eps_c=eps_c, grad=grad, loss=loss)
            coef = self.score_gs * (1.0 - a_t) ** 0.5
            return update(eps_c, eps_u, grad, coef)
        if eng.share_prefix:     # both halves of the CFG batch are the same latents: the text-free prefix runs once
            eps2 = eng.forward(latents, t, text, down_residuals=down, mid_residual=mid, dup=True)
        else:
            lat2 = latents.expand(2, -1, -1, -1, -1) if V == 1 else torch.cat([latents, latents], 0)
            eps2 = eng.forward(lat2, t, text, down_residuals=down, mid_residual=mid)
        T1 = eps2.shape[0] // 2
        if aux is not None:
            aux.update(eps_u=eps2[:T1], eps_c=eps2[T1:])
        return update(eps2[T1:], eps2[:T1], None, 0.0)

    def sample(self, latents, text, rep, progress=None, ctrl=None):
        rep_dev = self.engine.prepare_representation(rep)
        for i in range(len(self.timesteps)):
            latents = self.step(latents, i, text, rep_dev, ctrl=ctrl)
            if progress is not None:
                progress(i)
        return latents


def sample_interleaved(samplers, jobs, streams=None, add_noise_step=400, ctrl=None, on_step=None):
    """Several independent videos in flight on one GPU (examples are the unit of parallelism).

    `jobs[k] = (latents, text [2,77,768], reference_video_latents, extraction_noise)` - or a LIST of such tuples: V videos
    batched into one launch sequence on that lane (the result is then [V, 4, F, H, W]) - runs on `samplers[k]` / `streams[k]`
    (one sampler per lane: each owns its hipGraphs and static buffers); step i of every job is issued before step i + 1
    of any, so the kernels of the lanes interleave on the device and fill each other's tails.  Results are bit-identical
    to running the jobs one after the other (tools/concurrency_check.py) under the same `ops.set_gemm_share` setting
    (callers that keep 2 lanes busy set it to 2 once, before any graph is captured: the GEMM tile / split-K choice then
    targets half of the CUs per launch, +3.6 % videos/min).  `on_step(k, i, enter)` is called around every
    step inside the lane's stream context (bench.py records its events there).  `streams=None` issues everything on the
    current stream (same issue order, no overlap; what the host-simulator tests use).  Returns the final latents per job."""
    n = len(jobs)
    if n == 0:
        return []
    if n > len(samplers) or (streams is not None and n > len(streams)):
        raise ValueError("sample_interleaved: %d jobs for %d samplers / %s streams" % (
            n, len(samplers), "no" if streams is None else len(streams)))
    import contextlib
    lane = (lambda k: contextlib.nullcontext()) if streams is None else (lambda k: torch.cuda.stream(streams[k]))
    if streams is not None:
        first = jobs[0][0] if isinstance(jobs[0], list) else jobs[0]
        cur = torch.cuda.current_stream(first[0].device)
        for st in streams[:n]:
            st.wait_stream(cur)
    xs, reps, texts = [None] * n, [None] * n, [None] * n
    for k, job in enumerate(jobs):
        vids = job if isinstance(job, list) else [job]       # a LIST of videos on one lane = one batched launch sequence
        with lane(k):
            rr = [samplers[k].extract(vid, noise, text[0:1], add_noise_step=add_noise_step, ctrl=ctrl) for (_, text, vid, noise) in vids]
            reps[k] = samplers[k].engine.prepare_representation(rr if len(vids) > 1 else rr[0])
            xs[k] = vids[0][0] if len(vids) == 1 else torch.cat([v[0] for v in vids], 0)
            texts[k] = vids[0][1] if len(vids) == 1 else torch.cat([v[1][0:1] for v in vids] + [v[1][1:2] for v in vids], 0)
    for i in range(len(samplers[0].timesteps)):
        for k in range(n):
            with lane(k):
                if on_step is not None:
                    on_step(k, i, True)
                xs[k] = samplers[k].step(xs[k], i, texts[k], reps[k], ctrl=ctrl)
                if on_step is not None:
                    on_step(k, i, False)
    if streams is not None:
        for st in streams[:n]:
            cur.wait_stream(st)
    return xs
