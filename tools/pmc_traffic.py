"""HBM traffic per launch of the GEMM / conv kernels at the config-2 (B = 2) shapes, from rocprofv3 PMC passes.

  workload:  rocprofv3 --pmc FETCH_SIZE  --output-format csv -d OUT/FETCH_SIZE -- python tools/pmc_traffic.py run OUT
             rocprofv3 --pmc WRITE_SIZE  --output-format csv -d OUT/WRITE_SIZE -- python tools/pmc_traffic.py run OUT
  table:     python tools/pmc_traffic.py table OUT > profiles/hbm_traffic_per_shape.json   (+ a markdown table on stderr)
  PMC_LANES=3 in the environment of `run`: the tile choice of three videos in flight (default 1 = the roofline probe's regime);
  PMC_BATCH=5: V = 5 videos batched per lane (V times the rows of every launch), the packing of the round-6 bench

Separate passes, counters only (MI355X_MICROARCH.md: FETCH_SIZE takes 3 of the 4 TCC slots).  Units: KiB; on gfx950 FETCH_SIZE
reports half of a wide coalesced read, so traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes.  The library chooses the kernel
(cfg = 0); `run` records which structure ran for every launch (mc_gemm_last_kernel) so that `table` can pair the counter rows -
in dispatch order - with shapes and name them exactly as bench.py's roofline rows do."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, ".")
F2 = 32
SHAPES = [   # mode, M, N, K, geglu, residual, geometry
    (0, 32768, 1920, 640, False, False, None), (0, 32768, 640, 640, False, True, None), (0, 32768, 5120, 640, True, False, None),
    (0, 32768, 640, 2560, False, True, None), (0, 8192, 3840, 1280, False, False, None), (0, 8192, 1280, 1280, False, True, None),
    (0, 8192, 10240, 1280, True, False, None), (0, 131072, 320, 1280, False, True, None),
    (0, 131072, 960, 320, False, False, None), (0, 131072, 320, 320, False, True, None), (0, 131072, 2560, 320, True, False, None),
    (1, F2 * 4096, 320, 2880, False, True, (64, 64, 64, 64)), (1, F2 * 4096, 320, 5760, False, False, (64, 64, 64, 64)),
    (1, F2 * 1024, 640, 5760, False, True, (32, 32, 32, 32)), (1, F2 * 1024, 640, 11520, False, False, (32, 32, 32, 32)),
]
REPS = 2


def run(out):
    import torch
    from motionclone_amd import lib, ops
    from motionclone_amd.probe import _gemm_name
    dev = torch.device("cuda:0")
    # the tile choice depends on how many launch sequences the caller keeps in flight (ops.set_gemm_share): PMC_LANES = 1 is the
    # regime of bench.py's roofline probe, 3 the timed region's
    ops.set_gemm_share(int(os.environ.get("PMC_LANES", "1")))
    ops.TILELOOP_COL_OUTER = os.environ.get("PMC_COL_OUTER", "0") == "1"      # A/B: tile order of the persistent loop

    def r(*s, sc=1.0):
        return (torch.randn(*s, device=dev) * sc).half()
    order = []
    vb = int(os.environ.get("PMC_BATCH", "1"))     # videos batched per lane: V times the rows (frames) of every launch
    for mode, M, N, K, geglu, res, geom in SHAPES:
        M *= vb
        if mode == 0:
            x, kw = r(M, K), {}
        else:
            Hs, Ws, Ho, Wo = geom
            x, kw = r(M // (Ho * Wo) * Hs * Ws, K // 9), dict(mode=mode, geom=geom, m_out=M)
        w = r(N, K, sc=0.02)
        R = r(M, N) if res else None
        for _ in range(REPS):
            ops.gemm(x, w, residual=R, geglu=geglu, splits=1, **kw)   # split-K launches are two dispatches: not paired here
            order.append(dict(kernel=_gemm_name(lib.load().mc_gemm_last_kernel(), mode), shape=[mode, M, N, K, geglu, res]))
        torch.cuda.synchronize()
    json.dump(order, open(os.path.join(out, "order.json"), "w"))
    print("launched", len(order))


def table(out):
    order = json.load(open(os.path.join(out, "order.json")))
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True)
        rows = [r for r in csv.DictReader(open(f[0])) if "gemm" in r["Kernel_Name"] and "splitk_reduce" not in r["Kernel_Name"]
                and r["Counter_Name"] == c]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        vals[c] = rows
    # split-K launches are two dispatches (main + reduce): only single-dispatch launches are paired here
    n = len(order)
    assert all(len(v) == n for v in vals.values()), ({k: len(v) for k, v in vals.items()}, n)
    agg = {}
    for i, o in enumerate(order):
        key = (o["kernel"], tuple(o["shape"]))
        t = (2.0 * float(vals["FETCH_SIZE"][i]["Counter_Value"]) + float(vals["WRITE_SIZE"][i]["Counter_Value"])) * 1024.0
        agg.setdefault(key, []).append(t)
    res = []
    print("| kernel | mode M N K | HBM MB per launch | algorithmic MB | ratio |\n|---|---|---|---|---|", file=sys.stderr)
    for (k, sh), ts in agg.items():
        mode, M, N, K, geglu, resid = sh
        g = [g for (m_, M_, N_, K_, gg, rr, g) in SHAPES if (m_, N_, K_, gg, rr) == (mode, N, K, geglu, resid) and M % M_ == 0][0]
        rows_in = M if mode == 0 else (M // (g[2] * g[3])) * g[0] * g[1]
        k_in = K if mode == 0 else K // 9
        alg = 2.0 * (rows_in * k_in + N * K + M * (N // 2 if geglu else N) * (2 if resid else 1))
        t = sum(ts) / len(ts)
        res.append(dict(kernel=k, shape=list(sh), traffic_bytes=t, algorithmic_bytes=alg))
        print("| `%s` | %d %d %d %d%s%s | %.1f | %.1f | %.2f |" % (k, mode, M, N, K, " geglu" if geglu else "", " +R" if resid else "",
                                                                 t / 1e6, alg / 1e6, t / alg), file=sys.stderr)
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    {"run": run, "table": table}[sys.argv[1]](sys.argv[2])
