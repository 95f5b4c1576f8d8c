// C entry points of the GEMM / implicit-conv family (include/mc_kernels.h) and the choice of kernel structure:
//   gemm4.hip  streaming, A-stationary (short-K Linear layers of the 64x64 level, K = 320)
//   gemm5.hip  256x320 tiles, 8 waves, 4-stage ring + staggered LDS-DMA + wave-private epilogue (round 3: whatever fills the
//              256 CUs with 256x320 tiles and is not split-K)
//   gemm3.hip  256x320 / 128x320 tiles, 8 / 4 waves, 2-stage loop (split-K, 128x320 tiles, odd shapes)
//   gemm2.hip  128x128 / 64x64 tiles (small problems of the 16x16 / 8x8 levels)
// All kernels address their operands through 32-bit buffer descriptors (hardware range check = zero fill for conv
// padding and tails), so an operand must stay below 2 GiB: larger problems are cut into row ranges here.
#include "gemm_params.hpp"

namespace mc {
int gemm2_dispatch(const GemmParams& p, int mode, int small_tile, int deep, size_t rowsA, hipStream_t stream);   // gemm2.hip
int gemm3_dispatch(const GemmParams& p, int mode, int cfg, size_t rowsA, hipStream_t stream);                    // gemm3.hip
int gemm4_dispatch(const GemmParams& p, int nsplit, hipStream_t stream, const G4Norm* norm = nullptr);          // gemm4.hip
int gemm4_check(const GemmParams& p, const G4Norm* norm);                                                        // gemm4.hip
int gemm5_dispatch(const GemmParams& p, int mode, int var, size_t rowsA, hipStream_t stream);                    // gemm5.hip
int gemm6_dispatch(const GemmParams& p, int var, int mode, uint32_t* ctr, float* slabs, int max_wg, hipStream_t stream);   // gemm6.hip
size_t gemm6_slab_bytes();
size_t gemm6_counter_bytes();
int gn_partial_launch(const void* a, int lda, int ctot, int frames, int hw, float* partial, hipStream_t stream);   // norm.hip
}  // namespace mc

using namespace mc;

#ifndef MC_GEMM5_2WG_DEFAULT
#define MC_GEMM5_2WG_DEFAULT -1    // measured choice of the product library (profiles/r04_gemm5_2wg.md): by `share`
#endif

extern "C" int mc_gn_nchunk(int hw);   // norm.hip

#ifdef MC_TOOLS
// TOOLS BUILD ONLY: timing experiments that drop parts of a GEMM kernel (GemmParams::dbg); results are garbage.
static int g_gemm_debug = 0;
static float* g_gemm_debug_buf = nullptr;
extern "C" int mc_gemm_debug(int bits) {
    g_gemm_debug = bits;
    return 0;
}
extern "C" int mc_gemm_debug_buffer(void* buf) {   // device buffer for in-kernel cycle stamps (bit 16)
    g_gemm_debug_buf = (float*)buf;
    return 0;
}
#endif

// `share` (mc_gemm_f16 flags bits 20-21, mc_gemm_splitk_plan mode bits 8-9): the caller keeps 2^share independent launch
// sequences in flight on separate streams (sampler.sample_interleaved), so one launch only has to fill 1 / 2^share of the
// 256 CUs: the "enough workgroups" thresholds of the tile and split-K choice scale down with it, which keeps more work on the
// efficient 256x320 tiles and saves split-K round trips (measured with two videos in flight: +3.6 % videos/min for share 1,
// +1 % for share 2; with one video in flight share 1 costs 5 %).
static inline long fill_of(long full, int share) { return full >> share; }

// PROFILING ONLY (bench.py / motionclone_amd/probe.py): which kernel structure the calling thread's last mc_gemm_f16 /
// mc_gemm_splitk_f16 call used - 2 / 20 = gemm2 128x128 / 64x64, 31..35 = gemm3 geometry 1..5, 4 = gemm4, 51 / 54 = gemm5 with
// 256- / 128-row tiles; + 100 for a split-K call.  Lets the roofline table name kernels without restating the choice below.
static thread_local int g_last_kernel = 0;
extern "C" int mc_gemm_last_kernel(void) { return g_last_kernel; }

static int gemm_one(GemmParams p, int mode, int tile, int deep, int big_cfg, int nsplit, int share, hipStream_t s) {
    const int M = p.M, N = p.N;
    const size_t rowsA = mode == DENSE ? (size_t)M : (size_t)(M / (p.Ho * p.Wo)) * p.Hs * p.Ws;
    if (big_cfg == 10) {
        g_last_kernel = 4;
        return mode == DENSE ? gemm4_dispatch(p, nsplit, s) : MC_ERR_UNSUPPORTED;
    }
    if (p.gn_partial && (big_cfg == 11 || big_cfg == 15) && !tile && !deep) {   // forced geometry (tests): 256- / 128-row tiles
        const int rc5 = gemm5_dispatch(p, mode, big_cfg - 11, rowsA, s);
        g_last_kernel = big_cfg == 15 ? 54 : 51;
        return rc5 == MC_OK ? (big_cfg == 11 ? 64 : 32) : rc5;
    }
    if (big_cfg >= 11) {   // 11 = gemm5, 12-14 = schedule experiments, 15 = 128-row tiles
        g_last_kernel = big_cfg == 15 ? 54 : 51;
        return gemm5_dispatch(p, mode, big_cfg - 11, rowsA, s);
    }
    if (big_cfg == 7) {    // gemm5 with 256 x 256 tiles, FOUR waves, 128 x 128 wave tiles (round 6)
        g_last_kernel = 57;
        return gemm5_dispatch(p, mode, 7, rowsA, s);
    }
    if (big_cfg == 8) {    // gemm5 with 256 x 320 tiles, FOUR waves (128 x 160 wave tiles, one wave per SIMD: round 6)
        g_last_kernel = 58;
        return gemm5_dispatch(p, mode, 6, rowsA, s);
    }
    if (big_cfg == 9) {    // gemm5 with 256 x 160 tiles, 4 waves, two workgroups per CU
        g_last_kernel = 56;
        return gemm5_dispatch(p, mode, 5, rowsA, s);
    }
    if (p.gn_partial && (big_cfg || tile || deep)) return MC_ERR_UNSUPPORTED;   // statistics: the library's own choice only
    static const int no_g5 = MC_ENV_INT("MC_NO_GEMM5", 0);     // A/B, tools build only
    static const int g5_var = MC_ENV_INT("MC_GEMM5_VAR", 0);   // A/B, tools build only
    const bool automatic = !big_cfg && !tile && !deep;   // an explicit cfg = 1 still means gemm3 (tests, A/B tools)
    bool onewave_g3 = false;
    if (!big_cfg && !tile && !deep) {
        // measured on MI355X (profiles/r02_gemm4_microbench.md): the streaming kernel wins on the K = 320 Linear layers once
        // the problem has >= 256 row blocks of work; the 256x320 / 128x320 tiles win wherever they still fill the 256 CUs;
        // smaller problems stay on the 128x128 / 64x64 tiles
        static const int no_g4 = MC_ENV_INT("MC_NO_GEMM4", 0);   // diagnosis, tools build only
        if (!no_g4 && mode == DENSE && p.K == 320 && !p.A2 && (M >= 98304 || (M >= 32768 && N >= 640))) {
            if (p.gn_partial) return MC_ERR_UNSUPPORTED;     // (the streaming kernel's epilogue leaves no statistics)
            int rc4 = gemm4_dispatch(p, 0, s);
            g_last_kernel = 4;
            if (rc4 != MC_ERR_UNSUPPORTED) return rc4;
        }
        if (N % 320 == 0) {
            long b1 = (long)((M + 255) / 256) * (N / 320), b4 = (long)((M + 127) / 128) * (N / 320);
            if (b1 >= fill_of(224, share)) big_cfg = 1;
            else if (b4 >= fill_of(192, share)) big_cfg = 4;
            // A/B only (MC_GEMM_ONEWAVE128=1 | 2): a launch of about one wave of 256-row tiles as 128-row tiles - 1: two rounds of
            // gemm5 tiles (measured: 33.0 vs 33.8 videos/min), 2: gemm3's two workgroups per CU (33.9 vs 35.0) - so that the CUs
            // fall out of lockstep between the k-loop and the epilogue's memory phase
            static const int onewave128 = MC_ENV_INT("MC_GEMM_ONEWAVE128", 0);
            if (onewave128 && big_cfg == 1 && mode == DENSE && b1 <= 320) {
                big_cfg = 4;
                onewave_g3 = onewave128 == 2;   // 2: on gemm3's geometry (two 4-wave workgroups per CU, free to fall out of step)
            }
        }
    }
    // 256 x 160 tiles, 4 waves, TWO workgroups per CU (gemm5.hip, round 4).  G5_2WG: 0 = never, 1 = dense launches of about
    // one wave of 256 x 320 tiles, 2 = every dense launch with K <= 1280, 3 = every dense launch, 4 = wide short-K launches
    // (tools build: MC_GEMM5_2WG)
    // Measured inside the step loop (profiles/r04_gemm5_2wg.md): ONE video in flight +1.0 ... +2.8 % (rule 4 / rule 3), THREE
    // in flight -0.3 ... -2 % (the other streams already fill the prologue / epilogue gaps, and the geometry moves 1.44x the
    // operand bytes per MFMA) - so the default (-1) takes rule 4 only when the caller keeps a single launch sequence in flight
    if (p.gn_partial) {
        // GroupNorm statistics of the output from the epilogue (mc_gemm_gnstats_f16): the one-pass gemm5 kernels only; the return
        // value is the chunk height the statistics were written with (64 = 256-row tiles, 32 = 128-row tiles)
        if (no_g5 || (big_cfg != 1 && big_cfg != 4)) return MC_ERR_UNSUPPORTED;
        const int rc5 = gemm5_dispatch(p, mode, big_cfg == 1 ? 0 : 4, rowsA, s);
        g_last_kernel = big_cfg == 1 ? 51 : 54;
        return rc5 == MC_OK ? (big_cfg == 1 ? 64 : 32) : rc5;
    }
    static const int two_wg_env = MC_ENV_INT("MC_GEMM5_2WG", MC_GEMM5_2WG_DEFAULT);
    const int two_wg = two_wg_env >= 0 ? two_wg_env : (share == 0 ? 4 : 0);
    if (big_cfg == 1 && automatic && !no_g5 && two_wg && mode == DENSE && N % 160 == 0) {
        const long b1 = (long)((M + 255) / 256) * (N / 320);
        // 4 = the shapes that gained inside the step loop (profiles/r04_gemm5_2wg.md): wide outputs (>= 12 column tiles of
        // 160), short K, enough rows for both workgroups of every CU
        const bool wide_short = N >= 1920 && p.K <= 1280 && M >= 8192;
        if (two_wg == 3 || (two_wg == 2 && p.K <= 1280) || (two_wg == 1 && b1 <= 320) || (two_wg == 4 && wide_short)) {
            int rc5 = gemm5_dispatch(p, mode, 5, rowsA, s);
            g_last_kernel = 56;
            if (rc5 != MC_ERR_UNSUPPORTED) return rc5;
        }
    }
    if (big_cfg == 1 && automatic && !no_g5) {
        int rc5 = gemm5_dispatch(p, mode, (mode == DENSE || mode == CONV_S1) && !p.epi ? g5_var : 0, rowsA, s);
        g_last_kernel = 51;
        if (rc5 != MC_ERR_UNSUPPORTED) return rc5;
    }
    // A/B only (MC_GEMM5_NO128=1): the 128-row launches back on gemm3's 4-wave 128x320 geometry, TWO workgroups per CU (one
    // tile's epilogue under the other's k-loop) - slower when a launch is repeated (43.4 vs 37.7 us), not yet compared inside the step loop
    static const int no_g5_128 = MC_ENV_INT("MC_GEMM5_NO128", 0);
    if (big_cfg == 4 && automatic && !no_g5 && !no_g5_128 && !onewave_g3) {   // 128-row tiles with 8 waves instead of gemm3's 4-wave 128x320 geometry
        int rc5 = gemm5_dispatch(p, mode, 4, rowsA, s);
        g_last_kernel = 54;
        if (rc5 != MC_ERR_UNSUPPORTED) return rc5;
    }
    if (big_cfg) {
        int rc3 = gemm3_dispatch(p, mode, big_cfg, rowsA, s);
        g_last_kernel = 30 + big_cfg;
        if (rc3 != MC_ERR_UNSUPPORTED) return rc3;
    }
    int small_tile = tile == 64;
    if (tile == 0) {   // fall to 64x64 tiles when 128x128 would leave most of the 256 CUs idle
        long big = (long)((M + 127) / 128) * ((N + 127) / 128);
        small_tile = big < fill_of(256, share);
    }
    g_last_kernel = small_tile ? 20 : 2;
    return gemm2_dispatch(p, mode, small_tile, deep, rowsA, s);
}

static int gemm_entry(const void* A, const void* A2, const void* W, void* C, const void* R,
                      const float* bias, int M, int N, int K, int lda, int lda2, int ldc, int ldr,
                      int c1, int ctot, int mode, int Hs, int Ws, int Ho, int Wo,
                      int rows_per_batch, float alpha, int flags, void* stream, float* gn_partial, int gn_hw) {
    const int tile = flags & 0xFF;        // 0 = auto, 64, 128
    const int epi = (flags & 0x200) ? 1 : 0;   // fused GEGLU epilogue (weights row-interleaved h/gate)
    const int deep = (flags & 0x400) ? 1 : 0;  // 3-stage LDS ring of the small-tile kernels
    const int big_cfg = (flags >> 12) & 0xF;   // 0 = automatic; 1..5 gemm3 geometries; 10 = gemm4
    const int nsplit = (flags >> 16) & 0xF;    // gemm4: workgroups per 256-row block (0 = automatic)
    const int share = (flags >> 20) & 0x3;     // 2^share launch sequences in flight (see fill_of)
    if (M <= 0 || N <= 0 || K <= 0) return MC_ERR_SHAPE;
    if (flags & 0x100) return MC_ERR_UNSUPPORTED;   // the first-generation kernel is no longer part of the library
    if (epi && (R || N % 8)) return MC_ERR_UNSUPPORTED;
    if (K % BK || N % 4 || ldc % 4 || (R && (ldr % 4))) return MC_ERR_SHAPE;
    if (lda % 8 || (A2 && lda2 % 8)) return MC_ERR_SHAPE;
    if (mode < 0 || mode > 4) return MC_ERR_UNSUPPORTED;
    if (mode == DENSE) ctot = K;
    if (ctot <= 0 || ctot % BK || c1 % BK || c1 > ctot) return MC_ERR_SHAPE;
    if (c1 < ctot && !A2) return MC_ERR_SHAPE;
    if (mode != DENSE) {
        if (K != 9 * ctot || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0) return MC_ERR_SHAPE;
        if (M % (Ho * Wo)) return MC_ERR_SHAPE;
    }
    if (rows_per_batch <= 0) rows_per_batch = M;
    GemmParams p;
    p.A = (const half_t*)A; p.A2 = (const half_t*)A2; p.W = (const half_t*)W;
    p.C = (half_t*)C; p.R = (const half_t*)R; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.lda2 = lda2; p.ldc = ldc; p.ldr = ldr;
    p.c1 = c1; p.ctot = ctot; p.Hs = Hs; p.Ws = Ws; p.Ho = Ho; p.Wo = Wo;
    p.rows_per_batch = rows_per_batch; p.alpha = alpha; p.epi = epi;
#ifdef MC_TOOLS
    static const int dbg_env = MC_ENV_INT("MC_GEMM_DEBUG", 0);
    p.dbg = dbg_env | g_gemm_debug;
    p.ws = (p.dbg & 16) ? g_gemm_debug_buf : nullptr;
#else
    p.dbg = 0;
    p.ws = nullptr;
#endif
    p.splits = 1;
    p.s2_pad = (flags & 0x800) ? 0 : 1;
    p.gn_partial = gn_partial;
    p.gn_hw = gn_hw;
    hipStream_t s = (hipStream_t)stream;

    // 2 GiB descriptor limit: cut the problem into row ranges (whole frames for the conv modes).  With a per-batch bias
    // every range either holds whole batch entries or lies inside one, so its bias rows are a contiguous slice.
    static const size_t lim_env = (size_t)MC_ENV_INT("MC_GEMM_OPERAND_LIMIT", 0);   // tests (simulator build) only
    const size_t lim = lim_env ? lim_env : (size_t)0x7FFFFFF0u;
    const size_t out_cols = epi ? (size_t)N / 2 : (size_t)N;
    const size_t per_out_row = 2 * (size_t)std::max(std::max(ldc, ldr), (int)out_cols);
    size_t in_rows_per_unit, out_rows_per_unit, per_in_row = 2 * (size_t)std::max(lda, A2 ? lda2 : 0);
    if (mode == DENSE) {
        in_rows_per_unit = out_rows_per_unit = 1;
    } else {
        in_rows_per_unit = (size_t)Hs * Ws;
        out_rows_per_unit = (size_t)Ho * Wo;
    }
    const size_t units = (size_t)M / out_rows_per_unit;
    const size_t bytes_per_unit = std::max(in_rows_per_unit * per_in_row, out_rows_per_unit * per_out_row);
    size_t units_per_call = units;
    if (units * bytes_per_unit > lim) {
        if (gn_partial) return MC_ERR_UNSUPPORTED;   // (statistics are laid out for ONE launch)
        units_per_call = lim / bytes_per_unit;
        if (units_per_call == 0) return MC_ERR_UNSUPPORTED;
        if (bias && (size_t)rows_per_batch < (size_t)M) {   // keep every range inside / aligned with the bias batches
            const size_t upb = (size_t)rows_per_batch / out_rows_per_unit;   // units per batch entry
            if (upb == 0 || (size_t)rows_per_batch % out_rows_per_unit) return MC_ERR_UNSUPPORTED;
            if (units_per_call >= upb) units_per_call = units_per_call / upb * upb;
            else while (upb % units_per_call) --units_per_call;
        }
    }
    const bool per_batch_bias = bias && (size_t)rows_per_batch < (size_t)M;
    if (per_batch_bias && units_per_call < units) {
        // feasibility of every range BEFORE anything is launched: a range either starts on a batch boundary (and then holds
        // whole entries or part of one) or lies inside one entry
        for (size_t u0 = 0; u0 < units; u0 += units_per_call) {
            const size_t out0 = u0 * out_rows_per_unit, rows = std::min(units_per_call, units - u0) * out_rows_per_unit;
            const size_t in_entry = out0 % rows_per_batch;
            if (in_entry && in_entry + rows > (size_t)rows_per_batch) return MC_ERR_UNSUPPORTED;
        }
    }
    for (size_t u0 = 0; u0 < units; u0 += units_per_call) {
        const size_t nu = std::min(units_per_call, units - u0);
        GemmParams q = p;
        const size_t in0 = u0 * in_rows_per_unit, out0 = u0 * out_rows_per_unit;
        q.M = (int)(nu * out_rows_per_unit);
        q.A = p.A + in0 * lda;
        if (A2) q.A2 = p.A2 + in0 * lda2;
        q.C = p.C + out0 * ldc;
        if (R) q.R = p.R + out0 * ldr;
        if (per_batch_bias) {
            q.bias = bias + (out0 / rows_per_batch) * (size_t)N;
            // inside one batch entry: a single bias row serves the whole range
            q.rows_per_batch = (out0 % rows_per_batch) ? q.M : rows_per_batch;
        } else {
            q.rows_per_batch = q.M;
        }
        int rc = gemm_one(q, mode, tile, deep, big_cfg, nsplit, share, s);
        if (rc != MC_OK) return rc;      // (with gn_partial: one launch, and its chunk height > 0 is the result)
    }
    return MC_OK;
}

extern "C" int mc_gemm_f16(const void* A, const void* A2, const void* W, void* C, const void* R,
                           const float* bias, int M, int N, int K, int lda, int lda2, int ldc, int ldr,
                           int c1, int ctot, int mode, int Hs, int Ws, int Ho, int Wo,
                           int rows_per_batch, float alpha, int flags, void* stream) {
    return gemm_entry(A, A2, W, C, R, bias, M, N, K, lda, lda2, ldc, ldr, c1, ctot, mode, Hs, Ws, Ho, Wo, rows_per_batch, alpha,
                      flags, stream, nullptr, 0);
}

// mc_gemm_f16 that ALSO leaves the GroupNorm(32) partial sums of its output (round 6): the statistics pass of the norm that reads
// C next - resnet.py:197 (norm2 after conv1), the next block's norm1 / Transformer3DModel.norm / the motion module's norm after
// conv2 + shortcut (resnet.py:203-213, attention.py:105, motion_module.py:145) - is written by the producing kernel's epilogue.
//   gn_partial: float[(M / gn_hw) * (gn_hw / 32) * 64] (mc_workspace_bytes_gemm_gnstats), gn_hw: tokens per frame.
// Returns the CHUNK HEIGHT the sums were written with (64 or 32 rows: pass gn_hw / that to mc_groupnorm_fwd_partial_f16), or
// MC_ERR_UNSUPPORTED (-2) with NOTHING launched when the library's own choice for this problem is not a one-pass gemm5 kernel,
// N / 32 is not 10 / 20 / 40 channels per group, or a 64-row (32-row) wave tile would straddle frames: call mc_gemm_f16 and
// mc_groupnorm_fwd_f16 instead.
extern "C" long mc_workspace_bytes_gemm_gnstats(int frames, int hw) {
    return (frames <= 0 || hw <= 0 || hw % 32) ? -1 : (long)sizeof(float) * frames * (hw / 32) * 64;
}
extern "C" int mc_gemm_gnstats_f16(const void* A, const void* A2, const void* W, void* C, const void* R,
                                   const float* bias, int M, int N, int K, int lda, int lda2, int ldc, int ldr,
                                   int c1, int ctot, int mode, int Hs, int Ws, int Ho, int Wo,
                                   int rows_per_batch, float alpha, int flags, float* gn_partial, int gn_hw, void* stream) {
    if (!gn_partial || gn_hw <= 0 || (flags & 0x200)) return MC_ERR_UNSUPPORTED;
    return gemm_entry(A, A2, W, C, R, bias, M, N, K, lda, lda2, ldc, ldr, c1, ctot, mode, Hs, Ws, Ho, Wo, rows_per_batch, alpha,
                      flags, stream, gn_partial, gn_hw);
}

// ---- persistent tile loop (round 6, gemm6.hip) ---------------------------------------------------------------------------
// C[M,N] = alpha * A[M,K] . W[N,K]^T + bias + R (or the fused GEGLU epilogue) with ONE workgroup per CU walking the 256x320
// tiles: the operand ring runs through tile boundaries, so the three-stage prologue burst is paid once per launch instead
// of once per tile.  Same arithmetic as mc_gemm_f16's 256x320 kernel (bit-identical outputs unless a tile is cut, below).
//   workspace / ws_bytes: >= mc_workspace_bytes_gemm_tileloop(0) bytes that are ZERO when the kernel starts: the per-XCD tile
//     counters of the dynamic tile order and the hand-over flags of stream-K.  The kernel leaves them zero again, so one
//     zero-initialised block serves every launch of a stream (and every replay of a captured graph) without a memset;
//     launches that may run CONCURRENTLY need separate blocks.  workspace == null: static tile order (workgroup b walks
//     tiles b, b + grid, ...).
//   partials / partial_bytes (flags 0x2, stream-K): mc_workspace_bytes_gemm_tileloop(1) bytes of scratch (any contents).  The
//     k-stages of every XCD's tile list are dealt evenly to its workgroups; where a range ends inside a tile the tile is cut
//     along k, the later pieces leave their fp32 sums in `partials` and the workgroup that owns the tile's first stages adds
//     them in k order and runs the epilogue - what split-K + reduce did in two launches, without the quantisation of whole
//     tiles per workgroup.  A cut tile's sum is (first piece) + (second) + ...: deterministic, not bit-identical to the
//     one-chain sum of an uncut tile.
//   flags: 0x200 fused GEGLU (as mc_gemm_f16); 0x1 = do not assume that stores and loads retire in issue order (A/B);
//     0x2 = stream-K; bits 16-23: cap on the number of workgroups / 8 (0 = one per CU) - tests.
// MC_ERR_UNSUPPORTED: shapes outside the kernel (N % 8, K < 256, fewer than 8 row tiles without stream-K, two-source A,
// per-batch bias, operands >= 2 GiB) - call mc_gemm_f16.
extern "C" long mc_workspace_bytes_gemm_tileloop(int which) {
    return which == 0 ? (long)gemm6_counter_bytes() : which == 1 ? (long)gemm6_slab_bytes() : -1;
}

extern "C" int mc_gemm_tileloop_f16(const void* A, const void* A2, const void* W, void* C, const void* R, const float* bias,
                                    int M, int N, int K, int lda, int lda2, int ldc, int ldr, int c1, int rows_per_batch,
                                    float alpha, int flags, void* workspace, size_t ws_bytes, void* partials,
                                    size_t partial_bytes, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !W || !C) return MC_ERR_SHAPE;
    const int epi = (flags & 0x200) ? 1 : 0;
    if (epi && (R || N % 16)) return MC_ERR_UNSUPPORTED;
    if (K % BK || N % 4 || ldc % 4 || (R && (ldr % 4))) return MC_ERR_SHAPE;
    if (lda % 8 || (A2 && lda2 % 8)) return MC_ERR_SHAPE;
    if (c1 <= 0 || c1 > K || c1 % BK || (c1 < K && !A2)) return MC_ERR_SHAPE;
    if (workspace && (ws_bytes < gemm6_counter_bytes() || ((uintptr_t)workspace & 3))) return MC_ERR_SHAPE;
    const bool sk = (flags & 0x2) != 0;
    if (sk && (!workspace || !partials || partial_bytes < gemm6_slab_bytes() || ((uintptr_t)partials & 15))) return MC_ERR_SHAPE;
    if (rows_per_batch <= 0) rows_per_batch = M;
    GemmParams p;
    p.A = (const half_t*)A; p.A2 = (const half_t*)A2; p.W = (const half_t*)W;
    p.C = (half_t*)C; p.R = (const half_t*)R; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.lda2 = lda2; p.ldc = ldc; p.ldr = ldr;
    p.c1 = c1; p.ctot = K; p.Hs = p.Ws = p.Ho = p.Wo = 0;
    p.rows_per_batch = rows_per_batch; p.alpha = alpha; p.epi = epi; p.s2_pad = 1;
    p.ws = nullptr; p.splits = 1; p.dbg = 0;
    g_last_kernel = sk ? 62 : 61;
    return gemm6_dispatch(p, flags & 5, sk ? 2 : (workspace ? 1 : 0), (uint32_t*)workspace, (float*)partials,
                          ((flags >> 16) & 0xFF) * 8, (hipStream_t)stream);
}

// ---- norm + GEMM in one launch (round 4) ------------------------------------------------------------------------------
// C[M,N] = norm(A[M,320]) W[N,320]^T + bias with the normalisation applied in registers inside the streaming kernel
// (gemm4.hip): kind 1 = LayerNorm (+ temporal position table), kind 2 = GroupNorm(32) without activation (one more launch:
// the per-chunk partial sums).  MC_ERR_UNSUPPORTED = outside that kernel's shapes: the caller runs norm and GEMM separately.
extern "C" int mc_norm_gemm_f16(const void* A, const void* W, void* C, const float* bias, int M, int N, int K, int lda,
                                int ldc, int kind, const float* gamma, const float* beta, const float* pe, int hw,
                                int nframes_pe, float eps, float* stats, float* partial, int flags, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !W || !C || !gamma || !beta) return MC_ERR_SHAPE;
    if (kind != 1 && kind != 2) return MC_ERR_UNSUPPORTED;
    if (K != 320 || N % 32 || (lda & 7) || (ldc & 7)) return MC_ERR_UNSUPPORTED;
    const int epi = (flags & 0x200) ? 1 : 0;
    if (epi && N % 64) return MC_ERR_UNSUPPORTED;
    GemmParams p;
    p.A = (const half_t*)A; p.A2 = nullptr; p.W = (const half_t*)W; p.C = (half_t*)C; p.R = nullptr; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.lda2 = 0; p.ldc = ldc; p.ldr = 0; p.c1 = K; p.ctot = K;
    p.Hs = p.Ws = p.Ho = p.Wo = 0; p.rows_per_batch = M; p.alpha = 1.0f; p.epi = epi; p.s2_pad = 1;
    p.ws = nullptr; p.splits = 1; p.dbg = 0;
    G4Norm np{};
    np.kind = kind; np.gamma = gamma; np.beta = beta; np.pe = kind == 1 ? pe : nullptr; np.hw = hw;
    np.nframes_pe = nframes_pe; np.eps = eps; np.stats = stats;
    hipStream_t s = (hipStream_t)stream;
    if (kind == 2) {
        if (hw <= 0 || M % hw || hw % 256 || !partial || !stats) return MC_ERR_UNSUPPORTED;
        // flags bits 16-23 (round 6): `partial` already HOLDS the per-chunk sums, written with that many chunks per frame by the
        // kernel that produced A (mc_gemm_gnstats_f16): no statistics pass here
        const int ready_chunks = (flags >> 16) & 0xFF;
        np.partial = partial;
        np.nchunk = ready_chunks ? ready_chunks : mc_gn_nchunk(hw);
        np.gn_n = (float)hw * (K / 32);
        // nothing is launched for a problem the fused kernel will refuse (2 GiB descriptor limits, alignment): the caller's
        // fallback (GroupNorm, then GEMM) then runs the partial pass exactly once
        if (gemm4_check(p, &np) != MC_OK) return MC_ERR_UNSUPPORTED;
        int rc = ready_chunks ? MC_OK : gn_partial_launch(A, lda, K, M / hw, hw, partial, s);
        if (rc != MC_OK) return rc;
    }
    g_last_kernel = 4;
    return gemm4_dispatch(p, 0, s, &np);
}

// ---- split-K ------------------------------------------------------------------------------------------------------
namespace mc {
// out[m][n..n+3] = sum_s ws[s][m][n..] + bias[m / rows_per_batch][n..] + R[m][n..]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* ws, int splits, half_t* C, int ldc,
                                                            const half_t* R, int ldr, const float* bias, int M, int N,
                                                            int rows_per_batch) {
    const int nv = N / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)M * nv) return;
    const int m = (int)(idx / nv), n = (int)(idx - (long)m * nv) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(ws + (size_t)m * N + n);
    for (int s = 1; s < splits; ++s) {
        f32x4 x = *reinterpret_cast<const f32x4*>(ws + ((size_t)s * M + m) * N + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += x[e];
    }
    if (bias) {
        f32x4 b = *reinterpret_cast<const f32x4*>(bias + (size_t)(m / rows_per_batch) * N + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += b[e];
    }
    if (R) {
        half4_t r = ld4(R + (size_t)m * ldr + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
    }
    half4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = to_half(v[e]);
    st4(C + (size_t)m * ldc + n, o);
}
}  // namespace mc

// How many K ranges mc_gemm_splitk_f16 should be given for this problem: returns 1 (= use mc_gemm_f16) or
// splits | (geometry << 8), geometry = the gemm3 cfg to pass in flags bits 12-15.  Policy, measured on
// MI355X: the 8x8 / 16x16-level 3x3 convs (M <= 4096 rows, K = 11520 or 23040) leave 128x320 tiles on a fraction of
// the CUs or fall to 64x64 tiles at ~300 TFLOP/s; splitting K over up to 8 workgroups per tile fills the chip with
// the efficient geometry and costs one fp32 round trip of the (small) output (8x8 level: 143 -> 95 us at B = 2,
// 123 -> 60 us at B = 1; 16x16 level at B = 1: 189 -> 153 us).
extern "C" int mc_gemm_splitk_plan(int M, int N, int K, int mode) {
    const int share = (mode >> 8) & 0x3;
    if (N % 320 || K < 2304 || M <= 0) return 1;
    const int nk = K / BK;
    long t1 = (long)((M + 255) / 256) * (N / 320), t4 = (long)((M + 127) / 128) * (N / 320);
    const long fill = fill_of(256, share);
    if (t1 >= fill_of(224, share)) return 1;       // 256x320 tiles already fill (their share of) the chip
    int s, cfg;
    if (t1 >= fill_of(64, share)) {                // 64..223 big tiles: keep the efficient geometry, 2-4 K ranges
        s = (int)(fill / t1);
        cfg = 1;
    } else {                                       // fewer: 128x320 tiles, up to 8 K ranges
        if (t4 >= fill) return 1;
        s = (int)(fill / t4);
        cfg = 4;
    }
    if (s > 8) s = 8;
    while (s > 1 && nk / s < 8) --s;
    return s < 2 ? 1 : (s | (cfg << 8));
}

// Same contract as mc_gemm_f16 (no GEGLU epilogue), K split into `splits` ranges.  ws: fp32 workspace of
// splits * M * N elements.  flags bits 12-15 choose the gemm3 geometry (default 4: 128x320 tiles).
extern "C" int mc_gemm_splitk_f16(const void* A, const void* A2, const void* W, void* C, const void* R,
                                  const float* bias, int M, int N, int K, int lda, int lda2, int ldc, int ldr, int c1,
                                  int ctot, int mode, int Hs, int Ws, int Ho, int Wo, int rows_per_batch, float alpha,
                                  int flags, float* ws, int splits, void* stream) {
    int cfg = (flags >> 12) & 0xF;
    if (!cfg) cfg = 4;
    if (M <= 0 || N <= 0 || K <= 0 || !ws || splits < 1) return MC_ERR_SHAPE;
    if (flags & 0x200) return MC_ERR_UNSUPPORTED;
    if (cfg > 5) return MC_ERR_UNSUPPORTED;
    if (K % BK || N % 4 || ldc % 4 || (R && (ldr % 4))) return MC_ERR_SHAPE;
    if (lda % 8 || (A2 && lda2 % 8)) return MC_ERR_SHAPE;
    if (mode < 0 || mode > 4) return MC_ERR_UNSUPPORTED;
    if (mode == DENSE) ctot = K;
    if (ctot <= 0 || ctot % BK || c1 % BK || c1 > ctot) return MC_ERR_SHAPE;
    if (c1 < ctot && !A2) return MC_ERR_SHAPE;
    if (mode != DENSE) {
        if (K != 9 * ctot || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0) return MC_ERR_SHAPE;
        if (M % (Ho * Wo)) return MC_ERR_SHAPE;
    }
    if (splits > K / BK) splits = K / BK;
    if (rows_per_batch <= 0) rows_per_batch = M;
    GemmParams p;
    p.A = (const half_t*)A; p.A2 = (const half_t*)A2; p.W = (const half_t*)W;
    p.C = (half_t*)C; p.R = nullptr; p.bias = nullptr;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.lda2 = lda2; p.ldc = ldc; p.ldr = ldr;
    p.c1 = c1; p.ctot = ctot; p.Hs = Hs; p.Ws = Ws; p.Ho = Ho; p.Wo = Wo;
    p.rows_per_batch = rows_per_batch; p.alpha = alpha; p.epi = 0; p.dbg = 0;
    p.ws = ws; p.splits = splits;
    p.s2_pad = (flags & 0x800) ? 0 : 1;
    hipStream_t s = (hipStream_t)stream;
    size_t rowsA = mode == DENSE ? (size_t)M : (size_t)(M / (Ho * Wo)) * Hs * Ws;
    // gemm5 (ring kernel, 256- or 128-row tiles): partial sums in accumulator-native slabs + its own reduce / epilogue pass
    static const int no_g5 = MC_ENV_INT("MC_NO_GEMM5", 0);   // A/B, tools build only
    if (!no_g5 && splits > 1 && (cfg == 1 || cfg == 4) && !(flags & 0x1000000)) {
        GemmParams q = p;
        q.R = (const half_t*)R; q.bias = bias;
        int rc5 = gemm5_dispatch(q, mode, cfg == 4 ? 4 : 0, rowsA, s);
        g_last_kernel = cfg == 4 ? 154 : 151;
        if (rc5 != MC_ERR_UNSUPPORTED) return rc5;
    }
    g_last_kernel = 130 + cfg;
    int rc = gemm3_dispatch(p, mode, cfg, rowsA, s);
    if (rc != MC_OK) return rc;
    long nthr = (long)M * (N / 4);
    MC_LAUNCH(splitk_reduce_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, ws, splits, (half_t*)C, ldc,
              (const half_t*)R, ldr, bias, M, N, rows_per_batch);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}
