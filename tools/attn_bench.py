"""Spatial attention at the shapes of config 2 (16f x 512^2, 8 heads; CFG doubles the frame batch) on cuda:0 -> JSON lines.
Run once per setting of the library's switches (read once per process): `MC_ATTN_RING=0 MC_ATTN_TAG=old python
tools/attn_bench.py [--fwd-only]`."""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _toolslib  # noqa: E401,E702,F401  (tools build of the library: MC_* switches / debug hooks)
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from motionclone_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


tag = os.environ.get("MC_ATTN_TAG") or ("xcd=" + os.environ.get("MC_ATTN_XCD", "default"))
fwd_only = "--fwd-only" in sys.argv
for frames in (16, 32):
    for (name, N, d) in [("l0", 4096, 40), ("l1", 1024, 80), ("l2", 256, 160)]:
        C = 8 * d
        qkv = (torch.randn(frames * N, 3 * C, device=dev) * 0.5).half()
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        fl = 4.0 * frames * 8 * N * N * d
        ms = timeit(lambda: ops.attn_fwd(q, k, v, N, N, 8, d, frames))
        o, lse = ops.attn_fwd(q, k, v, N, N, 8, d, frames)
        do = (torch.randn(frames * N, C, device=dev)).half()
        dqkv = torch.empty_like(qkv)
        msb = float("nan") if fwd_only else timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, N, N, 8, d, frames, dq=dqkv[:, :C],
                                                                          dk=dqkv[:, C:2 * C], dv=dqkv[:, 2 * C:]), iters=10)
        print(json.dumps(dict(cfg=tag, level=name, frames=frames, fwd_ms=round(ms, 4), fwd_tflops=round(fl / ms / 1e9, 1),
                              bwd_ms=round(msb, 4), bwd_tflops=round(2.5 * fl / msb / 1e9, 1))), flush=True)

# cross-attention (77 text keys shared by the 16 frames of a video): forward + dQ
for frames in (16, 32):
    for (name, N, d) in [("x0", 4096, 40), ("x1", 1024, 80), ("x2", 256, 160)]:
        C = 8 * d
        q = (torch.randn(frames * N, C, device=dev) * 0.5).half()
        kv = (torch.randn(frames // 16 * 77, 2 * C, device=dev) * 0.5).half()
        k, v = kv[:, :C], kv[:, C:]
        ms = timeit(lambda: ops.attn_fwd(q, k, v, N, 77, 8, d, frames, kv_bdiv=16))
        o, lse = ops.attn_fwd(q, k, v, N, 77, 8, d, frames, kv_bdiv=16)
        do = (torch.randn(frames * N, C, device=dev)).half()
        dq = torch.empty_like(q)
        msb = float("nan") if fwd_only else timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, N, 77, 8, d, frames, kv_bdiv=16,
                                                                          dq=dq, need_dkv=False), iters=10)
        gb = 4.0 * frames * N * C   # q in, o out
        print(json.dumps(dict(cfg=tag, level=name, frames=frames, fwd_ms=round(ms, 4), fwd_gbps=round(gb / ms / 1e6, 1),
                              bwd_ms=round(msb, 4), bwd_gbps=round(2 * gb / msb / 1e6, 1))), flush=True)
