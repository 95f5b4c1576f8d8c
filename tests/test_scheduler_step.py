"""schedule_customized_step (motionclone_functions.py:285-409) with every branch, through mc_ddim_step_general_f16:
the oracle restatement is pinned to the reference's own function where the reference tree exists (CPU container), and
the kernel path is compared with the oracle on the host simulator and on the GPU."""
import itertools
import types

import pytest
import torch

from motionclone_amd.scheduler import DDIMSchedulerState
from motionclone_amd.utils import motionclone_functions as mf
from oracle import guidance_ref as G
from oracle import reference_shim as shim

SHAPE = (2, 4, 3, 8, 8)
CASES = [dict(prediction_type=p, clip_sample=c, eta=e, use_clipped_model_output=u)
         for p, c, e, u in itertools.product(("epsilon", "sample", "v_prediction"), (False, True), (0.0, 0.7), (False, True))]


def _tensors(seed=5):
    g = torch.Generator().manual_seed(seed)
    sample = torch.randn(SHAPE, generator=g)
    mo = torch.randn(SHAPE, generator=g) * 0.8
    score = torch.randn(SHAPE, generator=g) * 0.3
    noise = torch.randn(SHAPE, generator=g)
    return [t.half().float() for t in (sample, mo, score, noise)]


def _oracle(case, step_index, sample, mo, score, noise, sched, **kw):
    return G.ddim_step_general(sched.alphas_cumprod, sched.final_alpha_cumprod, sched.timesteps, step_index, mo, sample,
                               prediction_type=case["prediction_type"], clip_sample=case["clip_sample"],
                               clip_sample_range=1.0, eta=case["eta"],
                               use_clipped_model_output=case["use_clipped_model_output"],
                               variance_noise=noise if case["eta"] > 0 else None, score=score, **kw)


def _scheduler(case, dev):
    s = DDIMSchedulerState(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1,
                           clip_sample=case["clip_sample"], prediction_type=case["prediction_type"])
    s.customized_step = mf.schedule_customized_step.__get__(s)
    s.customized_set_timesteps = mf.schedule_set_timesteps.__get__(s)
    s.customized_set_timesteps(6, 3, 0.4, device=dev, timestep_spacing_type="uneven")
    return s


def _rel(a, b):
    return ((a.float().cpu() - b).norm() / b.norm().clamp_min(1e-20)).item()


@pytest.mark.skipif(not shim.available(), reason="reference tree not present")
@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(str(v) for v in c.values()))
def test_oracle_restatement_matches_the_reference_function(case):
    shim.install()
    import motionclone.utils.motionclone_functions as ref      # the reference's module (shim.install put it first on sys.path)
    assert ref.__file__.startswith(shim.REFERENCE_ROOT)
    sample, mo, score, noise = _tensors()
    sch = shim.DDIMScheduler(beta_start=0.00085, beta_end=0.012, steps_offset=1, clip_sample=case["clip_sample"],
                             prediction_type=case["prediction_type"])
    sch.variance_type = "fixed_small"
    sch.customized_step = ref.schedule_customized_step.__get__(sch)
    sch.customized_set_timesteps = ref.schedule_set_timesteps.__get__(sch)
    sch.customized_set_timesteps(6, 3, 0.4, device="cpu", timestep_spacing_type="uneven")
    for step_index in (0, 5):
        kw = dict(eta=case["eta"], use_clipped_model_output=case["use_clipped_model_output"],
                  variance_noise=noise if case["eta"] > 0 else None)
        got = sch.customized_step(mo.clone(), step_index, sample, score=score, guidance_scale=0.6, **kw)
        want = _oracle(case, step_index, sample, mo, score, noise, sch, guidance_scale=0.6)
        for a, b in zip(got[:2], want[:2]):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-5)
        assert abs(float(got[2]) - float(want[2])) < 1e-7
        # score on a batch subset, and the early return
        got = sch.customized_step(mo.clone(), step_index, sample, score=score[1:2], guidance_scale=0.6, indices=[1], **kw)
        want = _oracle(dict(case), step_index, sample, mo, score[1:2], noise, sch, guidance_scale=0.6, indices=[1])
        assert torch.allclose(got[0], want[0], rtol=1e-5, atol=1e-5)
        got = sch.customized_step(mo.clone(), step_index, sample, score=score, return_middle=True, **kw)
        want = _oracle(case, step_index, sample, mo, score, noise, sch, return_middle=True)
        assert torch.allclose(got[0], want[0], rtol=1e-5, atol=1e-5) and torch.allclose(got[3], want[3], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(str(v) for v in c.values()))
def test_kernel_path_matches_oracle(backend, case):
    dev = backend
    sample, mo, score, noise = _tensors()
    sch = _scheduler(case, dev)
    h = lambda t: t.to(dev).half()
    TOL = 3e-3      # fp16 storage of inputs / outputs, fp32 arithmetic inside the kernel
    for step_index in (0, 5):
        kw = dict(eta=case["eta"], use_clipped_model_output=case["use_clipped_model_output"],
                  variance_noise=h(noise) if case["eta"] > 0 else None)
        prev, x0, a_prev = sch.customized_step(h(mo), step_index, h(sample), score=h(score), guidance_scale=0.6, **kw)
        want = _oracle(case, step_index, sample, mo, score, noise, sch, guidance_scale=0.6)
        assert prev.dtype == torch.float16 and prev.shape == SHAPE
        assert _rel(prev, want[0]) < TOL and _rel(x0, want[1]) < TOL
        assert abs(float(a_prev) - float(want[2])) < 1e-7
        (only_prev,) = sch.customized_step(h(mo), step_index, h(sample), score=h(score), guidance_scale=0.6,
                                           return_dict=False, **kw)
        assert torch.equal(only_prev, prev)
        # guidance_scale 0 and score None are the un-guided update
        p0 = sch.customized_step(h(mo), step_index, h(sample), score=h(score), guidance_scale=0.0, **kw)[0]
        p1 = sch.customized_step(h(mo), step_index, h(sample), **kw)[0]
        assert torch.equal(p0, p1)
        assert _rel(p1, _oracle(case, step_index, sample, mo, None, noise, sch)[0]) < TOL
        # batch subset
        pi = sch.customized_step(h(mo), step_index, h(sample), score=h(score[1:2]), guidance_scale=0.6, indices=[1], **kw)[0]
        wi = _oracle(case, step_index, sample, mo, score[1:2], noise, sch, guidance_scale=0.6, indices=[1])[0]
        assert _rel(pi, wi) < TOL and torch.equal(pi[0], p1[0])
        # return_middle
        eps, a_t, a_p, x0m = sch.customized_step(h(mo), step_index, h(sample), score=h(score), return_middle=True, **kw)
        wm = _oracle(case, step_index, sample, mo, score, noise, sch, return_middle=True)
        assert _rel(eps, wm[0]) < TOL and _rel(x0m, wm[3]) < TOL and abs(float(a_t) - float(wm[1])) < 1e-7


def test_noise_is_drawn_like_randn_tensor_and_argument_errors(emu_device):
    dev = emu_device
    case = dict(prediction_type="epsilon", clip_sample=False, eta=1.0, use_clipped_model_output=False)
    sample, mo, score, noise = _tensors()
    sch = _scheduler(case, dev)
    g = torch.Generator().manual_seed(77)
    got = sch.customized_step(mo.half(), 1, sample.half(), eta=1.0, generator=g)[0]
    drawn = torch.randn(SHAPE, generator=torch.Generator().manual_seed(77), dtype=torch.float16)
    want = _oracle(case, 1, sample, mo, None, drawn.float(), sch)[0]
    assert _rel(got, want) < 3e-3
    with pytest.raises(ValueError, match="Cannot pass both generator and variance_noise"):
        sch.customized_step(mo.half(), 1, sample.half(), eta=1.0, generator=g, variance_noise=noise.half())
    sch.num_inference_steps = None
    with pytest.raises(ValueError, match="Number of inference steps is 'None'"):
        sch.customized_step(mo.half(), 1, sample.half())
    with pytest.raises(NotImplementedError):
        DDIMSchedulerState(thresholding=True)
    with pytest.raises(ValueError, match="prediction_type"):
        DDIMSchedulerState(prediction_type="other")
