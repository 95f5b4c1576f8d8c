"""Per-launch measurement of the C ABI (bench.py's roofline table; tools).

`LaunchProbe` wraps `lib.call`, so EVERY kernel entry point the engine issues is bracketed by a pair of HIP events on the
launch stream (the pinned / current torch stream - the one the library launches on), and classified into a kernel family
with its ALGORITHMIC work taken from the call's own arguments:

  * GEMM / conv: 2 M N K flop; bytes = every operand once (activations, weights, residual, output); the kernel structure
    that ran is asked from the library (mc_gemm_last_kernel), not re-derived here;
  * spatial attention: 4 Nq Nk d flop per (frame, head) forward, 2.5x that for the backward pair;
  * everything else (temporal attention, GroupNorm, LayerNorm, GEGLU, adds, the fused DDIM update): bytes of every operand
    once - those kernels are HBM-bound.

A row's BINDING roof is the one it is closer to: fraction of the dense fp16 MFMA peak (2.5 PFLOP/s) vs fraction of HBM
(8 TB/s spec; ~6.3 TB/s is what a copy reaches).  Events cost ~1 us of host time each: the probe runs on one eager video
outside the timed region, never inside it."""
import torch

from . import lib

PEAK_TFLOPS = 2500.0     # fp16 dense MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBPS = 8000.0   # spec; achievable ~6300

_GEMM_KERNELS = {2: "gemm2<128x128>", 20: "gemm2<64x64>", 4: "gemm4<K=320 streaming>", 51: "gemm5<256x320>",
                 54: "gemm5<128x320>", 56: "gemm5<256x160 x2 per CU>", 57: "gemm5<256x256, 4 waves>", 58: "gemm5<256x320, 4 waves>",
                 61: "gemm6<256x320 tile loop>", 62: "gemm6<256x320 tile loop, stream-K>"}
_MODES = ["DENSE", "CONV_S1", "CONV_S2", "CONV_UP", "TCONV_S2"]


def _gemm_name(code, mode):
    split = code >= 100
    code %= 100
    base = _GEMM_KERNELS.get(code, "gemm3<cfg %d>" % (code - 30))
    return "%s %s%s" % (base, _MODES[mode], " split-K + reduce" if split else "")


def _cost(name, a):
    """-> (family, flop, bytes, shape key) of one C-ABI call with positional arguments `a`; None = not a launch"""
    if name in ("mc_gemm_f16", "mc_gemm_splitk_f16", "mc_gemm_gnstats_f16"):   # (the same leading arguments)
        M, N, K, mode, flags = a[6], a[7], a[8], a[15], a[22]
        c1, ctot, Hs, Ws, Ho, Wo = a[13], a[14], a[16], a[17], a[18], a[19]
        geglu = bool(flags & 0x200)
        rows_in = M if mode == 0 else (M // max(1, Ho * Wo)) * Hs * Ws
        k_in = K if mode == 0 else ctot
        nout = N // 2 if geglu else N
        nbytes = 2.0 * (rows_in * k_in + N * K + M * nout * (2 if a[4] else 1))
        return ("gemm", 2.0 * M * N * K, nbytes, (mode, M, N, K, geglu, bool(a[4])))
    if name == "mc_gemm_tileloop_f16":
        # (A, A2, W, C, R, bias, M, N, K, lda, lda2, ldc, ldr, c1, rows_per_batch, alpha, flags, workspace, ws_bytes, partials,
        #  partial_bytes, stream)
        M, N, K, flags = a[6], a[7], a[8], a[16]
        geglu = bool(flags & 0x200)
        nout = N // 2 if geglu else N
        nbytes = 2.0 * (M * K + N * K + M * nout * (2 if a[4] else 1))
        return ("gemm6<256x320 tile loop%s> DENSE" % (", stream-K" if flags & 2 else ""), 2.0 * M * N * K, nbytes,
                (0, M, N, K, geglu, bool(a[4])))
    if name == "mc_norm_gemm_f16":
        # (A, W, C, bias, M, N, K, lda, ldc, kind, gamma, beta, pe, hw, nframes_pe, eps, stats, partial, flags, stream)
        M, N, K, kind, flags = a[4], a[5], a[6], a[9], a[18]
        nout = N // 2 if flags & 0x200 else N
        # algorithmic bytes of what it replaces and still has to move: A once (twice for GroupNorm: the statistics pass), W, C
        nbytes = 2.0 * (M * K * (2 if kind == 2 else 1) + N * K + M * nout)
        return ("gemm4<K=320 streaming> %s + DENSE" % ("LayerNorm" if kind == 1 else "GroupNorm"), 2.0 * M * N * K, nbytes,
                (kind, M, N, K, bool(flags & 0x200)))
    if name == "mc_attn_fwd_f16":
        Nq, Nk, heads, d, nb = a[9], a[10], a[11], a[12], a[13]
        fl = 4.0 * Nq * Nk * d * heads * nb
        kind = "self" if a[14] == 1 else "cross"
        return ("attn_fwd %s d=%d Nk=%d" % (kind, d, Nk), fl, 2.0 * heads * d * nb * (2 * Nq + 2 * Nk / a[14]), (Nq, Nk, d, nb))
    if name == "mc_attn_bwd_f16":
        Nq, Nk, heads, d, nb = a[18], a[19], a[20], a[21], a[22]
        fl = 10.0 * Nq * Nk * d * heads * nb
        kind = "self" if a[23] == 1 else "cross"
        return ("attn_bwd %s d=%d Nk=%d" % (kind, d, Nk), fl, 2.0 * heads * d * nb * (4 * Nq + 4 * Nk / a[23]), (Nq, Nk, d, nb))
    if name == "mc_tattn_fwd_f16":
        B, F, HW, heads, d = a[6], a[7], a[8], a[9], a[10]
        return ("tattn_fwd", 4.0 * B * HW * heads * F * F * d, 8.0 * B * F * HW * heads * d, (B, F, HW, d))
    if name == "mc_tattn_bwd_f16":
        B, F, HW, heads, d = a[13], a[14], a[15], a[16], a[17]
        return ("tattn_bwd", 10.0 * B * HW * heads * F * F * d, (14.0 if a[4] else 12.0) * B * F * HW * heads * d, (B, F, HW, d))
    if name in ("mc_tattn_top1_f16", "mc_tattn_prob_f16", "mc_tattn_loss_f16"):
        return ("tattn_readout", 0.0, 0.0, ())
    if name == "mc_groupnorm_stats_f16":
        ctot, frames, hw = a[5], a[6], a[7]
        return ("groupnorm_stats", 0.0, 2.0 * frames * hw * ctot, (frames * hw, ctot))
    if name == "mc_groupnorm_apply_f16":
        ctot, frames, hw = a[5], a[6], a[7]
        return ("groupnorm_apply", 0.0, 4.0 * frames * hw * ctot, (frames * hw, ctot))
    if name == "mc_groupnorm_fwd_f16":
        ctot, frames, hw = a[5], a[6], a[7]
        return ("groupnorm_fwd", 0.0, 6.0 * frames * hw * ctot, (frames * hw, ctot))
    if name == "mc_groupnorm_fwd_partial_f16":     # statistics from the producing GEMM's epilogue: one read, one write
        ctot, frames, hw = a[2], a[3], a[4]
        return ("groupnorm_fwd (statistics from the producer)", 0.0, 4.0 * frames * hw * ctot, (frames * hw, ctot))
    if name == "mc_groupnorm_bwd_f16":
        ctot, frames, hw = a[5], a[6], a[7]
        return ("groupnorm_bwd", 0.0, (8.0 if a[18] else 6.0) * frames * hw * ctot, (frames * hw, ctot))
    if name == "mc_layernorm_fwd_f16":
        M, C = a[10], a[11]
        return ("layernorm_fwd", 0.0, 4.0 * M * C, (M, C))
    if name == "mc_layernorm_bwd_f16":
        M, C = a[10], a[11]
        return ("layernorm_bwd", 0.0, (8.0 if a[6] else 6.0) * M * C, (M, C))
    if name == "mc_geglu_fwd_f16":
        M, D = a[4], a[5]
        return ("geglu_fwd", 0.0, 6.0 * M * D, (M, D))
    if name == "mc_geglu_bwd_f16":
        M, D = a[6], a[7]
        return ("geglu_bwd", 0.0, 10.0 * M * D, (M, D))
    if name == "mc_add_f16":
        M, C = a[6], a[7]
        return ("add", 0.0, 6.0 * M * C, (M, C))
    if name == "mc_sumpool2_f16":
        frames, H, W, C = a[4], a[5], a[6], a[7]
        return ("sumpool2", 0.0, 2.0 * frames * H * W * C * 5, (frames * H * W, C))
    if name.startswith("mc_workspace_bytes_") or name in ("mc_version", "mc_gemm_splitk_plan", "mc_gn_nchunk", "mc_gemm_debug",
                                                          "mc_gemm_last_kernel", "mc_attn_last_kernel", "mc_tattn_last_kernel", "mc_gemm_debug_buffer",
                                                          "mc_tattn_debug_buffer"):
        return None
    return (name[3:].replace("_f16", "").replace("_f32", ""), 0.0, 0.0, ())   # small elementwise / layout kernels


class LaunchProbe:
    def __init__(self):
        self.records = []     # (family, e0, e1, flop, bytes, shape)
        self.enabled = False
        self._orig = None

    def install(self):
        if self._orig is not None:
            return self
        self._orig = lib.call
        probe = self

        def call(name, *args):
            if not probe.enabled:
                return probe._orig(name, *args)
            c = _cost(name, args)
            if c is None:
                return probe._orig(name, *args)
            st = args[-1]     # every entry point takes the launch stream as its last argument
            ext = torch.cuda.ExternalStream(st) if st else torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ext)
            rc = probe._orig(name, *args)
            e1.record(ext)
            fam, fl, nb, shape = c
            if fam == "gemm":
                fam = _gemm_name(lib.load().mc_gemm_last_kernel(), shape[0])
            elif fam.startswith("attn_") and lib.load().mc_attn_last_kernel():
                fam = fam.replace("attn_fwd", "attn_fwd_ring").replace("attn_bwd", "attn_bwd_ring")   # which structure ran
            elif fam in ("tattn_fwd", "tattn_bwd") and lib.load().mc_tattn_last_kernel():
                fam += "_vec"
            probe.records.append((fam, e0, e1, fl, nb, shape))
            return rc
        lib.call = call
        # entry points that may answer "unsupported" (lib.try_call: mc_norm_gemm_f16) are bracketed the same way; a refused
        # call launched nothing and leaves no record
        self._orig_try = lib.try_call

        def try_call(name, *args):
            if not probe.enabled:
                return probe._orig_try(name, *args)
            n0 = len(probe.records)
            saved, probe._orig = probe._orig, probe._orig_try
            try:
                ok = call(name, *args)
            finally:
                probe._orig = saved
            if not ok:
                del probe.records[n0:]
            return ok
        lib.try_call = try_call
        # ... and the ones that answer with a count (lib.call_count: mc_gemm_gnstats_f16 -> chunk height, False = nothing launched)
        self._orig_count = lib.call_count

        def call_count(name, *args):
            if not probe.enabled:
                return probe._orig_count(name, *args)
            n0 = len(probe.records)
            saved, probe._orig = probe._orig, probe._orig_count
            try:
                rc = call(name, *args)
            finally:
                probe._orig = saved
            if rc is False:
                del probe.records[n0:]
            return rc
        lib.call_count = call_count
        return self

    def uninstall(self):
        if self._orig is not None:
            lib.call = self._orig
            lib.try_call = self._orig_try
            lib.call_count = self._orig_count
            self._orig = None

    def reset(self):
        self.records = []

    def by_shape(self):
        rows = {}
        for fam, e0, e1, fl, nb, shape in self.records:
            r = rows.setdefault((fam, shape), dict(launches=0, ms=0.0, flop=0.0, bytes=0.0))
            r["launches"] += 1
            r["ms"] += e0.elapsed_time(e1)
            r["flop"] += fl
            r["bytes"] += nb
        out = []
        for (fam, shape), r in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]):
            out.append(dict(kernel=fam, shape=list(shape), launches=r["launches"], ms=r["ms"],
                            avg_us=1e3 * r["ms"] / r["launches"], tflops=r["flop"] / r["ms"] / 1e9 if r["ms"] else 0.0,
                            alg_gbps=r["bytes"] / r["ms"] / 1e6 if r["ms"] else 0.0,
                            algorithmic_bytes_per_launch=r["bytes"] / r["launches"]))
        return out

    def summary(self, wall_s):
        """family -> roofline row; `wall_s` = wall time of the probed region (shares are of that)"""
        groups = {}
        for fam, e0, e1, fl, nb, _ in self.records:
            g = groups.setdefault(fam, dict(launches=0, ms=0.0, flop=0.0, bytes=0.0))
            g["launches"] += 1
            g["ms"] += e0.elapsed_time(e1)
            g["flop"] += fl
            g["bytes"] += nb
        rows = {}
        for fam, g in groups.items():
            if g["ms"] <= 0.0:
                continue
            tf = g["flop"] / g["ms"] / 1e9
            gbps = g["bytes"] / g["ms"] / 1e6
            f_mfma, f_hbm = tf / PEAK_TFLOPS, gbps / PEAK_HBM_GBPS
            mfma_bound = f_mfma >= f_hbm
            rows[fam] = dict(bound="mfma" if mfma_bound else "hbm",
                             achieved=tf if mfma_bound else gbps, peak=PEAK_TFLOPS if mfma_bound else PEAK_HBM_GBPS,
                             unit="TFLOP/s" if mfma_bound else "GB/s", frac=max(f_mfma, f_hbm),
                             frac_of_mfma_peak=f_mfma, frac_of_hbm_peak=f_hbm, launches=g["launches"],
                             avg_launch_us=1e3 * g["ms"] / g["launches"], flop_per_launch=g["flop"] / g["launches"],
                             algorithmic_bytes_per_launch=g["bytes"] / g["launches"], tflops=tf, algorithmic_gbps=gbps,
                             traffic=None, share_of_probe_video=g["ms"] / 1e3 / wall_s)
        return rows

    def covered(self, wall_s):
        return sum(e0.elapsed_time(e1) for _, e0, e1, _, _, _ in self.records) / 1e3 / wall_s
