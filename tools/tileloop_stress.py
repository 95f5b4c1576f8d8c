"""Stress of the persistent tile loop's dynamic tile order (mc_gemm_tileloop_f16) under UNEVEN load: two launch sequences with
their own counter blocks on two streams, a third stream that keeps random subsets of the CUs busy with other kernels, every
output compared bit for bit with gemm5's (computed alone, beforehand).  Any element that differs, any NaN left from the
pre-fill (= a tile nobody computed) and any counter word that is not zero afterwards is an error.

  python tools/tileloop_stress.py [--iters 40]
"""
import argparse
import json
import sys

import torch

sys.path.insert(0, ".")
from motionclone_amd import lib, ops  # noqa: E402

dev = torch.device("cuda:0")


def rnd(*shape, s=1.0, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(*shape, device=dev, generator=g) * s).half()


class Owner:
    pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    a = ap.parse_args()
    lib.load()
    shapes = [(32768, 5120, 640, True), (8192, 3840, 1280, False), (8192, 10240, 1280, True), (4096, 3840, 1280, False)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    noise = torch.cuda.Stream()
    owners = [Owner(), Owner()]
    ops.prepare_tile_counters(dev)
    bad_total = 0
    for (M, N, K, geglu) in shapes:
        x, w = rnd(M, K, seed=1), rnd(N, K, s=0.02, seed=2)
        bias = torch.randn(1, N, device=dev) * 0.1
        ref = ops.gemm(x, w, bias=bias, geglu=geglu, cfg=11)
        outs = [torch.empty_like(ref) for _ in streams]
        nx = [rnd(m, 640, seed=7 + i) for i, m in enumerate((2048, 8192, 30000))]
        nw = rnd(640, 640, s=0.02, seed=9)
        torch.cuda.synchronize()
        bad = 0
        for it in range(a.iters):
            with torch.cuda.stream(noise):      # other launch sequences: kernels of varying size that take CUs for a while
                for k in range(3):
                    ops.gemm(nx[(it + k) % 3], nw, cfg=11)
            for si, st in enumerate(streams):
                with torch.cuda.stream(st):
                    ops._PIN.owner = owners[si]
                    try:
                        outs[si].fill_(float("nan"))
                        r = ops.gemm_tileloop(x, w, bias=bias, geglu=geglu, out=outs[si], dynamic=True)
                        assert r is not None
                    finally:
                        ops._PIN.owner = None
            if it % 4 == 3:
                torch.cuda.synchronize()
                for o in outs:
                    bad += int((o != ref).sum().item())
        torch.cuda.synchronize()
        for o in outs:
            bad += int((o != ref).sum().item())
        dirty = sum(int(v.count_nonzero().item()) for v in ops._tile_slabs.values())
        print(json.dumps(dict(M=M, N=N, K=K, geglu=geglu, launches=2 * a.iters, unequal_or_missing_elements=bad,
                              counter_words_nonzero_after=dirty, counter_blocks=len(ops._tile_blocks))), flush=True)
        bad_total += bad + dirty
    print("STRESS", "OK" if bad_total == 0 else "FAILED")
    sys.exit(0 if bad_total == 0 else 1)


if __name__ == "__main__":
    main()
