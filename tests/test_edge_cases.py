"""Ragged / boundary shapes and error paths of the C ABI, on the host simulator (and the GPU when marked): sizes that are
not multiples of any tile, single rows, maximum frame counts, argument validation (status codes -> RuntimeError)."""
import pytest
import torch
import torch.nn.functional as Fn
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from motionclone_amd import ops

COMMON = dict(deadline=None, max_examples=12, suppress_health_check=[HealthCheck.function_scoped_fixture])


def rnd(shape, seed, scale=1.0):
    return (torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale).half()


@settings(**COMMON)
@given(M=st.integers(1, 300), N4=st.integers(1, 90), K64=st.integers(1, 4), tile=st.sampled_from([0, 64, 128]),
       cfg=st.sampled_from([0, 0, 1, 3, 4, 5]), use_res=st.booleans())
def test_gemm_ragged_shapes(backend, M, N4, K64, tile, cfg, use_res):
    dev = backend
    N, K = 4 * N4, 64 * K64
    if cfg:
        tile = 0
    a, w = rnd((M, K), 1).to(dev), rnd((N, K), 2, 0.1).to(dev)
    bias = torch.randn(1, N, generator=torch.Generator().manual_seed(3)).to(dev)
    res = rnd((M, N), 4).to(dev) if use_res else None
    out = ops.gemm(a, w, bias=bias, residual=res, tile=tile, cfg=cfg)
    ref = a.float() @ w.float().t() + bias + (res.float() if use_res else 0)
    assert (out.float() - ref).abs().max() <= 2e-2 + 5e-3 * ref.abs().max()


@settings(**COMMON)
@given(Nq=st.integers(1, 140), Nk=st.integers(1, 140), d=st.sampled_from([8, 16, 40, 64, 80]), heads=st.integers(1, 3),
       nb=st.integers(1, 2))
def test_attention_ragged_shapes(backend, Nq, Nk, d, heads, nb):
    dev = backend
    C = heads * d
    q = rnd((nb * Nq, C), 1, 0.7).to(dev)
    kv = rnd((nb * Nk, 2 * C), 2, 0.7).to(dev)
    o, lse = ops.attn_fwd(q, kv[:, :C], kv[:, C:], Nq, Nk, heads, d, nb)
    Q = q.float().reshape(nb, Nq, heads, d).permute(0, 2, 1, 3)
    K = kv[:, :C].float().reshape(nb, Nk, heads, d).permute(0, 2, 1, 3)
    V = kv[:, C:].float().reshape(nb, Nk, heads, d).permute(0, 2, 1, 3)
    S = (Q @ K.transpose(-1, -2)) * d ** -0.5
    ref = (S.softmax(-1) @ V).permute(0, 2, 1, 3).reshape(nb * Nq, C)
    assert (o.float() - ref).abs().max() < 1e-2
    assert (lse - torch.logsumexp(S, -1)).abs().max() < 3e-3


@pytest.mark.parametrize("F_", [1, 2, 17, 32])
def test_temporal_attention_frame_counts(backend, F_):
    """1 frame (softmax of a single score), a non-multiple of the 16-wide tile, and the maximum 32"""
    dev = backend
    B, HW, heads, d = 1, 5, 2, 16
    C = heads * d
    qkv = rnd((B * F_ * HW, 3 * C), 7, 0.8).to(dev)
    o = ops.tattn_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, F_, HW, heads, d)
    t = qkv.float().reshape(B, F_, HW, 3, heads, d).permute(3, 0, 2, 4, 1, 5)          # [3, B, HW, heads, F, d]
    ref = (((t[0] @ t[1].transpose(-1, -2)) * d ** -0.5).softmax(-1) @ t[2])           # [B, HW, heads, F, d]
    ref = ref.permute(0, 3, 1, 2, 4).reshape(B * F_ * HW, C)
    assert (o.float() - ref).abs().max() < 1e-2


def test_argument_validation(backend):
    dev = backend
    a, w = rnd((8, 64), 1).to(dev), rnd((8, 64), 2).to(dev)
    with pytest.raises(RuntimeError, match="bad shape"):
        ops.gemm(a[:, :32], w[:, :32])                       # K not a multiple of 64
    with pytest.raises(RuntimeError, match="bad shape"):
        ops.gemm(a, rnd((6, 64), 3).to(dev))                 # N not a multiple of 4
    with pytest.raises(RuntimeError, match="unsupported"):
        ops.gemm(a, w, residual=rnd((8, 8), 4).to(dev), geglu=True)
    with pytest.raises(RuntimeError):
        ops.tattn_fwd(a[:, :16], a[:, 16:32], a[:, 32:48], 1, 33, 1, 1, 16)      # more than 32 frames
    with pytest.raises(RuntimeError, match="bad shape"):
        ops.softmax_rows_(rnd((4, 12), 5).to(dev))           # columns not a multiple of 8
    with pytest.raises(AssertionError):
        ops.gemm(a.float(), w)                               # fp32 activations are refused by the wrapper


def test_groupnorm_single_frame_tiny_grid(backend):
    dev = backend
    x = rnd((7, 64), 9).to(dev)                              # hw = 7: fewer rows than a chunk
    g, b = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    st_ = ops.gn_stats(x, None, 1, 7, 1e-5)
    y = ops.gn_apply(x, None, st_, g, b, False, 1, 7)
    ref = Fn.group_norm(x.float().t().reshape(1, 64, 7), 32).reshape(64, 7).t()
    assert (y.float() - ref).abs().max() < 1e-2


def test_gemm_row_range_splitting_below_the_descriptor_limit():
    """operands above the 2 GiB buffer-descriptor limit are cut into row ranges by the C entry point (whole frames for the
    conv modes, bias batches kept aligned); exercised in a child process with the limit lowered to 40 KiB"""
    import os
    import subprocess
    import sys
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from motionclone_amd import lib, build, ops
lib.use_library_for_tests(build.build_emu())
import torch.nn.functional as Fn
g = torch.Generator().manual_seed(0)
M, N, K = 400, 72, 128
a = (torch.randn(M, K, generator=g) * 0.5).half(); w = (torch.randn(N, K, generator=g) * 0.1).half()
bias = torch.randn(4, N, generator=g); res = torch.randn(M, N, generator=g).half()
out = ops.gemm(a, w, bias=bias, residual=res, rows_per_batch=100)
ref = a.float() @ w.float().t() + bias.repeat_interleave(100, 0) + res.float()
assert (out.float() - ref).abs().max() < 3e-2, (out.float() - ref).abs().max()
# the limit below ONE batch entry (200 rows x 256 B > 40 KiB): ranges of 100 rows lie inside an entry, the second range of
# every entry starts off a batch boundary and takes that entry's bias row
bias2 = torch.randn(2, N, generator=g)
out2 = ops.gemm(a, w, bias=bias2, residual=res, rows_per_batch=200)
ref2 = a.float() @ w.float().t() + bias2.repeat_interleave(200, 0) + res.float()
assert (out2.float() - ref2).abs().max() < 3e-2, (out2.float() - ref2).abs().max()
NF, C, H, W = 6, 64, 8, 8
x = (torch.randn(NF, C, H, W, generator=g)).half(); wc = (torch.randn(72, C, 3, 3, generator=g) * 0.05).half()
xc = x.permute(0, 2, 3, 1).reshape(-1, C).contiguous()
tb = torch.randn(2, 72, generator=g)
wp = ops.pack_conv_k(wc.float().permute(0, 2, 3, 1).reshape(72, 9, C)).half()
o = ops.gemm(xc, wp, bias=tb, rows_per_batch=3 * H * W, mode=ops.CONV_S1, geom=(H, W, H, W), m_out=NF * H * W)
refc = Fn.conv2d(x.float(), wc.float(), padding=1) + tb.repeat_interleave(3, 0)[:, :, None, None]
got = o.float().reshape(NF, H, W, 72).permute(0, 3, 1, 2)
assert (got - refc).abs().max() < 3e-2, (got - refc).abs().max()
print("split ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MC_GEMM_OPERAND_LIMIT="40960")
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
    assert r.returncode == 0 and "split ok" in r.stdout, r.stdout[-3000:]
