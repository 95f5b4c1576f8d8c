"""A/B of the GroupNorm statistics from the producing GEMM's epilogue (round 6, mc_gemm_gnstats_f16 + mc_groupnorm_fwd_partial_f16)
against the statistics pass (mc_gemm_f16 + mc_groupnorm_fwd_f16), on the 3x3 convolutions and Linear layers whose output the
engine normalises next, AS THE ENGINE CALLS THEM (three videos in flight: share of 3 lanes).

  python tools/gn_epilogue_ab.py [--rounds 7] [--iters 10]

One JSON line per shape: median us of (GEMM, GroupNorm + SiLU) per arm, HIP events around --iters back-to-back pairs."""
import argparse
import json
import statistics
import sys

import torch

sys.path.insert(0, ".")
from motionclone_amd import lib, ops  # noqa: E402

dev = torch.device("cuda:0")
F2 = 32


def r(*shape, s=1.0, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(*shape, device=dev, generator=g) * s).half()


# name, mode, frames, hw, N, K, residual, per-sample bias, geom
SHAPES = [
    ("conv1_l0 320->320 +temb", 1, F2, 4096, 320, 2880, False, True, (64, 64, 64, 64)),
    ("conv2_l0 320->320 +R", 1, F2, 4096, 320, 2880, True, False, (64, 64, 64, 64)),
    ("conv1_l0 640->320 +temb", 1, F2, 4096, 320, 5760, False, True, (64, 64, 64, 64)),
    ("conv1_l1 640->640 +temb", 1, F2, 1024, 640, 5760, False, True, (32, 32, 32, 32)),
    ("conv2_l1 640->640 +R", 1, F2, 1024, 640, 5760, True, False, (32, 32, 32, 32)),
    ("conv1_l1 1280->640 +temb", 1, F2, 1024, 640, 11520, False, True, (32, 32, 32, 32)),
    ("conv1_l2 1280->1280 +temb", 1, F2, 256, 1280, 11520, False, True, (16, 16, 16, 16)),
    ("proj_out_l1 +R (dense, no tile loop)", 0, F2, 1024, 640, 640, True, False, None),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--lanes", type=int, default=3)
    a = ap.parse_args()
    lib.load()
    ops.set_gemm_share(a.lanes)
    ops.TILELOOP = False
    print(json.dumps(dict(library=lib.HIP_LIB_PATH, lanes=a.lanes)), flush=True)
    for name, mode, frames, hw, N, K, res, temb, geom in SHAPES:
        M = frames * hw
        if mode == 0:
            x = r(M, K, seed=1)
            kw = dict()
        else:
            x = r(M, K // 9, seed=1)
            kw = dict(mode=mode, geom=geom, m_out=M)
        w = r(N, K, s=0.02, seed=2)
        R = r(M, N, seed=3) if res else None
        bias, rpb = (torch.randn(2, N, device=dev), M // 2) if temb else (torch.randn(1, N, device=dev), 0)
        gamma, beta = torch.ones(N, device=dev), torch.zeros(N, device=dev)
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        y = torch.empty((M, N), dtype=torch.float16, device=dev)
        row = dict(shape=name, M=M, N=N, K=K)
        res_y = {}
        for arm in ("statistics pass", "epilogue"):
            ops.GN_FROM_EPILOGUE = arm == "epilogue"

            def fn():
                o, gp = ops.gemm(x, w, bias=bias, rows_per_batch=rpb, residual=R, out=out, gn_hw=hw, **kw)
                ops.gn_fwd(o, None, gamma, beta, True, frames, hw, 1e-5, out=y, gnp=gp)
                return gp
            gp = fn()
            torch.cuda.synchronize()
            ts = []
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(a.rounds):
                e0.record()
                for _ in range(a.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(1e3 * e0.elapsed_time(e1) / a.iters)
            row[arm] = dict(us=round(statistics.median(ts), 1), from_epilogue=gp is not None, kernel=lib.load().mc_gemm_last_kernel())
            res_y[arm] = y.clone()
        row["max_abs_diff_of_normalised_output"] = float((res_y["epilogue"].float() - res_y["statistics pass"].float()).abs().max())
        row["saved_us"] = round(row["statistics pass"]["us"] - row["epilogue"]["us"], 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
