"""HBM traffic of `LayerNorm / GroupNorm kernel + K = 320 GEMM` vs the one fused launch (mc_norm_gemm_f16), from rocprofv3 PMC passes:
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT/FETCH_SIZE -- python tools/pmc_norm_gemm.py run
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d OUT/WRITE_SIZE -- python tools/pmc_norm_gemm.py run
  python tools/pmc_norm_gemm.py table OUT
Launch order of `run`: for (LayerNorm + q|k|v N = 960, GroupNorm + proj_in N = 320) at M = 131072: [norm kernel(s), GEMM], [fused]."""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    from motionclone_amd import lib, ops
    lib.load()
    ops.NORM_GEMM_MIN_ROWS = 0
    dev = torch.device("cuda:0")
    M, hw = 131072, 4096
    x = (torch.randn(M, 320, device=dev)).half()
    gamma, beta = torch.rand(320, device=dev) + 0.5, torch.randn(320, device=dev) * 0.1
    for kind, N in ((1, 960), (2, 320)):
        w = (torch.randn(N, 320, device=dev) * 0.05).half()
        for _ in range(2):
            if kind == 1:
                n, _ = ops.layernorm_fwd(x, gamma, beta)
            else:
                n, _ = ops.gn_fwd(x, None, gamma, beta, False, M // hw, hw, 1e-6)
            ops.gemm(n, w)
            assert ops.norm_gemm(x, w, kind, gamma, beta, hw=hw, eps=1e-5 if kind == 1 else 1e-6) is not None
        torch.cuda.synchronize()


def table(out):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True)[0]
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void mc::", "")
            if "at::" in name or "elementwise" in name or "distribution" in name or "fill" in name.lower():
                continue
            d = tot.setdefault(name, dict(FETCH_SIZE=0.0, WRITE_SIZE=0.0, n=0))
            d[c] += float(r["Counter_Value"])
            if c == "FETCH_SIZE":
                d["n"] += 1
    print("| kernel | launches | MB read per launch (2 x FETCH_SIZE) | MB written per launch | total MB |\n|---|---|---|---|---|")
    for k, d in tot.items():
        rd, wr = 2.0 * d["FETCH_SIZE"] * 1024 / d["n"] / 1e6, d["WRITE_SIZE"] * 1024 / d["n"] / 1e6
        print("| `%s` | %d | %.1f | %.1f | %.1f |" % (k[:70], d["n"], rd, wr, rd + wr))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else table(sys.argv[2])
