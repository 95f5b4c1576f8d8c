"""mc_norm_gemm_f16 (norm inside the K = 320 streaming GEMM) against the two launches it replaces, on the level-0 shapes of
the config-2 / config-1 forward.  Interleaved rounds, buffers rotated through 6 copies (the operands of a launch are not the
ones the previous launch left in the caches).  One JSON line per case.
  python tools/norm_gemm_bench.py > gpurun_out/r04_norm_gemm.jsonl"""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motionclone_amd import lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib.load()
ops.NORM_GEMM_MIN_ROWS = 0
NCOPY = 6


def r(*shape, seed=0, s=1.0):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(*shape, device=dev, generator=g) * s).half()


def timeit(fn, rounds=5, iters=8):
    for _ in range(2):
        fn(0)
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / iters)
    return statistics.median(ts)


CASES = [  # name, kind, N, geglu, pe
    ("norm1 + qkv", 1, 960, False, False), ("norm2 + q", 1, 320, False, False), ("norm3 + ff1 geglu", 1, 2560, True, False),
    ("norm3 + ff1 (taped rows)", 1, 2560, False, False), ("temporal norm + pe + qkv", 1, 960, False, True),
    ("groupnorm + proj_in", 2, 320, False, False),
]
for M in (131072, 65536, 32768):
    hw = 4096 if M >= 65536 else 1024
    xs = [r(M, 320, seed=10 + i) for i in range(NCOPY)]
    gamma, beta = torch.rand(320, device=dev) + 0.5, torch.randn(320, device=dev) * 0.1
    pe = torch.randn(16, 320, device=dev)
    for name, kind, N, geglu, with_pe in CASES:
        w = r(N, 320, seed=3, s=0.05)
        bias = torch.randn(1, N, device=dev) * 0.1
        outs = [torch.empty((M, N // 2 if geglu else N), device=dev, dtype=torch.float16) for _ in range(NCOPY)]
        nbuf = [torch.empty((M, 320), device=dev, dtype=torch.float16) for _ in range(NCOPY)]

        def fused(i):
            assert ops.norm_gemm(xs[i % NCOPY], w, kind, gamma, beta, bias=bias, pe=pe if with_pe else None, hw=hw,
                                 geglu=geglu, out=outs[i % NCOPY]) is not None

        def split(i):
            x = xs[i % NCOPY]
            if kind == 1:
                n, _ = ops.layernorm_fwd(x, gamma, beta, pe=pe if with_pe else None, hw=hw, out=nbuf[i % NCOPY])
            else:
                n, _ = ops.gn_fwd(x, None, gamma, beta, False, M // hw, hw, 1e-6, out=nbuf[i % NCOPY])
            ops.gemm(n, w, bias=bias, geglu=geglu, out=outs[i % NCOPY])

        def norm_only(i):
            x = xs[i % NCOPY]
            if kind == 1:
                ops.layernorm_fwd(x, gamma, beta, pe=pe if with_pe else None, hw=hw, out=nbuf[i % NCOPY])
            else:
                ops.gn_fwd(x, None, gamma, beta, False, M // hw, hw, 1e-6, out=nbuf[i % NCOPY])
        tf, ts, tn = timeit(fused), timeit(split), timeit(norm_only)
        print(json.dumps(dict(case=name, M=M, N=N, fused_us=round(tf, 1), two_launches_us=round(ts, 1), norm_alone_us=round(tn, 1),
                              gemm_alone_us=round(ts - tn, 1), saved_us=round(ts - tf, 1))), flush=True)
        del outs, nbuf
