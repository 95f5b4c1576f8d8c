"""gemm4 (streaming short-K kernel) vs the tiled kernels on the level-0 Linear shapes: correctness against torch + timing."""
import json
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


g = torch.Generator(device=dev).manual_seed(0)
for (M, N, K, res, geglu) in [(131072, 320, 320, True, False), (131072, 320, 320, False, False), (131072, 960, 320, False, False),
                              (131072, 2560, 320, False, False), (131072, 2560, 320, False, True), (65536, 320, 320, False, False),
                              (65536, 1280, 320, False, False), (65536, 2560, 320, False, False), (16384, 320, 320, True, False)]:
    a = (torch.randn(M, K, device=dev, generator=g) * 0.5).half()
    w = (torch.randn(N, K, device=dev, generator=g) * 0.05).half()
    b = torch.randn(1, N, device=dev, generator=g)
    r = torch.randn(M, N, device=dev, generator=g).half() if res else None
    fl = 2.0 * M * N * K
    nout = N // 2 if geglu else N
    nbytes = 2.0 * (M * K + N * K + M * nout * (2 if res else 1))
    row = dict(M=M, N=N, K=K, res=res, geglu=geglu)
    if geglu:
        wi, bi = ops.interleave_geglu(w), ops.interleave_geglu(b[0]).unsqueeze(0).contiguous()
        base = ops.gemm(a, wi, bias=bi, geglu=True)
        ms = timeit(lambda: ops.gemm(a, wi, bias=bi, geglu=True))
        row.update(tiled_us=1e3 * ms, tiled_tf=fl / ms / 1e9)
        full = a[:4096].float() @ w.float().t() + b
        ref = (full[:, :N // 2] * torch.nn.functional.gelu(full[:, N // 2:])).half().float()
        for ns in (0, 1, 2, 4):
            out = ops.gemm(a, wi, bias=bi, geglu=True, cfg=10, nsplit=ns)
            err = (out[:4096].float() - ref).abs().max().item()
            same = (out.float() - base.float()).abs().max().item()
            ms4 = timeit(lambda: ops.gemm(a, wi, bias=bi, geglu=True, cfg=10, nsplit=ns))
            row["g4_ns%d" % ns] = dict(us=1e3 * ms4, tf=fl / ms4 / 1e9, tbps=nbytes / ms4 / 1e9, err_vs_torch=err, max_diff_vs_tiled=same)
    else:
        base = ops.gemm(a, w, bias=b, residual=r)
        ms = timeit(lambda: ops.gemm(a, w, bias=b, residual=r))
        row.update(tiled_us=1e3 * ms, tiled_tf=fl / ms / 1e9, tiled_tbps=nbytes / ms / 1e9)
        ref = (a[:4096].float() @ w.float().t() + b).half().float()
        if res:
            ref = (ref + r[:4096].float())
        for ns in (0, 1, 2, 4):
            try:
                out = ops.gemm(a, w, bias=b, residual=r, cfg=10, nsplit=ns)
            except RuntimeError as e:
                row["g4_ns%d" % ns] = str(e)
                continue
            err = (out[:4096].float() - ref).abs().max().item()
            same = (out.float() - base.float()).abs().max().item()
            ms4 = timeit(lambda: ops.gemm(a, w, bias=b, residual=r, cfg=10, nsplit=ns))
            row["g4_ns%d" % ns] = dict(us=1e3 * ms4, tf=fl / ms4 / 1e9, tbps=nbytes / ms4 / 1e9, err_vs_torch=err, max_diff_vs_tiled=same)
    print(json.dumps(row))

# pure write / copy bandwidth of this box (torch elementwise kernels)
x = torch.empty(335544320 // 2, dtype=torch.float16, device=dev)   # 335 MB
y = torch.empty_like(x)
ms = timeit(lambda: x.zero_())
print(json.dumps(dict(probe="zero_ 335 MB", us=1e3 * ms, write_tbps=x.numel() * 2 / ms / 1e9)))
ms = timeit(lambda: y.copy_(x))
print(json.dumps(dict(probe="copy_ 335 MB", us=1e3 * ms, total_tbps=2 * x.numel() * 2 / ms / 1e9)))
