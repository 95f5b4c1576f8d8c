"""K = 320 Linear layers of the 64x64 level (proj_out / to_out + residual, to_q|k|v, to_k|v of the motion modules): the streaming kernel
(gemm4, the library's choice) against the persistent tile loop (gemm6, forced), at V videos batched per lane.
  python tools/k320_ab.py [--batch 5]"""
import argparse
import json
import statistics
import sys

import torch

sys.path.insert(0, ".")
from motionclone_amd import lib, ops  # noqa: E402

dev = torch.device("cuda:0")


def r(*shape, s=1.0, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(*shape, device=dev, generator=g) * s).half()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=6)
    a = ap.parse_args()
    lib.load()
    ops.set_gemm_share(2)
    for name, N, res, geglu in (("to_out / proj_out +R", 320, True, False), ("q|k|v", 960, False, False), ("k|v", 640, False, False),
                                ("proj_in", 320, False, False), ("ff1 geglu", 2560, False, True)):
        M, K = a.batch * 131072, 320
        x, w = r(M, K, seed=1), r(N, K, s=0.03, seed=2)
        R = r(M, N, seed=3) if res else None
        bias = torch.randn(1, N, device=dev)
        outs = {k: torch.empty((M, N // 2 if geglu else N), dtype=torch.float16, device=dev) for k in ("streaming (gemm4)", "tile loop (gemm6)")}
        fns = {"streaming (gemm4)": lambda: ops.gemm(x, w, bias=bias, residual=R, geglu=geglu, out=outs["streaming (gemm4)"], tileloop=False),
               "tile loop (gemm6)": lambda: ops.gemm(x, w, bias=bias, residual=R, geglu=geglu, out=outs["tile loop (gemm6)"], tileloop=True)}
        kern = {}
        for k, f in fns.items():
            f()
            kern[k] = lib.load().mc_gemm_last_kernel()
        torch.cuda.synchronize()
        ts = {k: [] for k in fns}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(a.rounds):
            for k, f in fns.items():
                e0.record()
                for _ in range(a.iters):
                    f()
                e1.record()
                torch.cuda.synchronize()
                ts[k].append(1e3 * e0.elapsed_time(e1) / a.iters)
        nbytes = 2.0 * (M * K + N * K + M * (N // 2 if geglu else N) * (2 if res else 1))
        row = dict(shape=name, M=M, N=N, K=K, max_abs_diff=float((outs["streaming (gemm4)"].float() - outs["tile loop (gemm6)"].float()).abs().max()))
        for k in fns:
            us = statistics.median(ts[k])
            row[k] = dict(us=round(us, 1), TFLOPs=round(2.0 * M * N * K / us / 1e6), alg_TBps=round(nbytes / us / 1e6, 2), kernel=kern[k])
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
