// The 256x320 ring-kernel vocabulary shared by gemm5.hip (one tile per workgroup) and gemm6.hip (persistent tile loop):
// tile geometry, the XOR-swizzled LDS offset of a staged row, and the wave-private epilogue.
#pragma once
#include "gemm_params.hpp"
#include <type_traits>
#include <cstdlib>

namespace mc {

namespace g5 {
constexpr int BN = 320, NW = 8, NT = 512, TN = 5, BKT = 32, NS = 4;   // the 8-wave geometries (defaults of Tile<>)
constexpr int ROWB = 64;                 // bytes per staged row (32 halfs)
constexpr int RPI = 16;                  // rows moved by one LDS-DMA instruction
constexpr int RS = 336;                  // row pitch of the epilogue's fp16 image (bytes)
constexpr int RSG = 176;                 // same, fused GEGLU (80 outputs per row)
constexpr int STG = 32 * RS;             // one 32-row block of a wave's tile
// Geometries (wave tile = 32 TM rows x 160 columns of v_mfma_f32_32x32x16_f16, TN = 5 column blocks):
//   <256, 320, 8, 4>  8 waves as 4 x 2, wave tile 64 x 160, ring of 4 stages (144 KiB): ONE workgroup per CU - the default
//   <128, 320, 8, 4>  8 waves, wave tile 32 x 160: problems that do not fill the 256 CUs with 256-row tiles (16x16 / 8x8 levels)
//   <256, 320, 4, 4>  (round 6) 4 waves as 2 x 2, ONE wave per SIMD with the whole 512-register file: wave tile 128 x 160 -
//                     36 % fewer LDS fragment bytes per MFMA than the 64 x 160 wave tile (461 instead of 717), half the waves
//                     at every barrier; 320 accumulator registers per lane
//   <256, 256, 4, 4>  (round 6) 4 waves as 2 x 2, wave tile 128 x 128 (TM = 4, FOUR column blocks): the 256 accumulator registers
//                     are exactly the AGPR file, every VGPR is left to fragments and addresses; 0.5 LDS fragment reads per MFMA
//                     (0.7 for the 64 x 160 wave tile) and operands fetched once per 256 columns.  For the layers whose N is a
//                     multiple of 256 and wide (FeedForward's first Linear, q|k|v at 1280): the geometry of the vendor
//                     library's kernel on those shapes (profiles/r05_vendor_kernel_names.md)
//   <256, 160, 4, 3>  (round 4) 4 waves as 4 x 1, wave tile 64 x 160, ring of 3 stages (78 KiB): TWO workgroups per CU that
//                     run out of step - one tile's prologue / epilogue (residual read, stores) under the other's k-loop.  Same
//                     per-wave code as the default (TM = 2), 1.44x the operand bytes per MFMA (A is fetched once per 160
//                     columns instead of once per 320).  For the short-K launches of about one wave of tiles.
template <int BM, int BN_ = BN, int NW_ = NW, int NS_ = NS>
struct Tile {
    static constexpr int BNT = BN_, NWV = NW_, NSV = NS_, NTH = NW_ * 64;
    static constexpr int WCOL = BN_ % 160 == 0 ? 160 : 128;  // columns of a wave tile: 5 or 4 blocks of 32 (TNV)
    static constexpr int TNV = WCOL / 32;
    static constexpr int WNW = BN_ / WCOL, WMW = NW_ / WNW;  // waves along N / along M
    static constexpr int TM = BM / (32 * WMW);
    static constexpr int RA = BM / RPI / NW_;            // activation row groups per wave and stage
    static constexpr int WB = (BN_ / RPI) / NW_;         // weight row groups every wave moves per stage ...
    static constexpr int WX = (BN_ / RPI) % NW_;         // ... and the first WX waves one more
    static constexpr int STAGE = (BM + BN_) * ROWB;      // 36864 / 28672 / 26624
    static constexpr int A_BYTES = BM * ROWB;
    static constexpr int LB = RA + WB, LA = RA + WB + (WX > 0 ? 1 : 0); // LDS-DMA instructions per stage: waves >= WX / waves < WX
    static constexpr size_t SMEM = (size_t)NS_ * STAGE;
    static_assert(BN_ % WCOL == 0 && NW_ % WNW == 0 && BM % (32 * WMW) == 0 && BM % (RPI * NW_) == 0, "tile / wave grid");
    static constexpr int WXD = WX > 0 ? WX : 1;          // (divisor of `wave % WX` where no wave moves an extra group)
    static_assert(NW_ * (32 * RSG + 640) <= NS_ * STAGE, "epilogue images must fit the ring");
    static_assert(NS_ == 3 || NS_ == 4, "ring depth");
};

__device__ __forceinline__ int lds_off32(int row, int v) { return row * 64 + ((v ^ ((row >> 2) & 3)) << 4); }
}  // namespace g5

// Wave-private epilogue of one wave tile (32 TM rows x 160 columns at global (mw0, nw0)): bias / alpha in fp32 in the
// accumulator layout, fp16 through the wave's LDS image `stg` (32 rows, pitch 336 B), read back as whole 320-byte row segments
// and stored 16 bytes per lane with the residual added on the way.  No workgroup barrier.
template <int EPI, int TM>
__device__ __forceinline__ void g5_epilogue(const GemmParams& p, f32x16 (&acc)[g5::TN][TM], char* stg, int mw0, int nw0, int lane) {
    using namespace g5;
    const int l31 = lane & 31, lhi = lane >> 5;
    constexpr int SEGS = EPI == 1 ? 10 : 20;          // 16-byte segments per image row
    constexpr int PITCH = EPI == 1 ? RSG : RS;
    constexpr int RPI_OUT = 60 / SEGS;                // rows per read-back instruction (60 of 64 lanes)
    constexpr int NIT = (32 + RPI_OUT - 1) / RPI_OUT;
    const int seg = lane % SEGS, rsel = lane / SEGS;  // lanes 60..63: rsel == RPI_OUT -> idle
    const int ncol = EPI == 1 ? nw0 / 2 + seg * 8 : nw0 + seg * 8;   // first output column of the lane's segment
    const int nout = EPI == 1 ? p.N / 2 : p.N;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int mrow = mw0 + 32 * j;                // global row of image row 0
        // residual rows of this half: in flight while the accumulators are converted and transposed
        half8_t rres[NIT];
        if (EPI == 0 && p.R) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int r = it * RPI_OUT + rsel, m = mrow + r;
                if (rsel < RPI_OUT && r < 32 && m < p.M && ncol < nout) rres[it] = ld8(p.R + (size_t)m * p.ldr + ncol);
            }
        }
        // accumulators (lane: row l31, 4 consecutive columns per (i, q)) -> + bias, * alpha -> fp16 image
        const int mlane = mrow + l31;
        const float* brow = p.bias ? p.bias + (size_t)(min(mlane, p.M - 1) / p.rows_per_batch) * p.N : nullptr;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cl = 32 * i + 8 * q + 4 * lhi;   // column within the wave's 160
                const int n = nw0 + cl;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e] * p.alpha;
                if (brow && n < p.N) {
                    f32x4 b = *reinterpret_cast<const f32x4*>(brow + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += b[e];
                }
                if (EPI == 1) {   // fused GEGLU: columns are (h, gate) pairs
                    half2_t o;
                    o[0] = to_half(v[0] * gelu_f(v[1]));
                    o[1] = to_half(v[2] * gelu_f(v[3]));
                    *reinterpret_cast<half2_t*>(stg + l31 * PITCH + cl) = o;
                } else {
                    half4_t o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = to_half(v[e]);
                    *reinterpret_cast<half4_t*>(stg + l31 * PITCH + cl * 2) = o;
                }
            }
        }
        wave_lds_sync();
        // image rows -> global: 16 bytes per lane, whole row segments of the wave's columns
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = it * RPI_OUT + rsel, m = mrow + r;
            if (rsel < RPI_OUT && r < 32 && m < p.M && ncol < nout) {
                half8_t o = *reinterpret_cast<const half8_t*>(stg + r * PITCH + seg * 16);
                if (EPI == 0 && p.R) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = to_half((float)o[e] + (float)rres[it][e]);
                }
                st8(p.C + (size_t)m * p.ldc + ncol, o);
            }
        }
        wave_lds_sync();   // the image is rewritten by the next half
    }
}


// ---- the round-6 epilogue (gemm6.hip; gemm5.hip's one-pass kernels use it too) -------------------------------------------------
#ifdef MC_EMU
#define MC_SCHED_FENCE() ((void)0)
#else
#define MC_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// The lane index, recomputed where it is needed (two VALU instructions) instead of kept: nothing inside the k-loop reads it,
// so a lane index (or anything derived from it) that lives across the loop is spilled to scratch, and every reload is a
// scratch_load + s_waitcnt vmcnt(0) - a drain of the operand ring and of the epilogue's stores.  `opaque` keeps hipcc from
// hoisting the recomputation (and every address that depends on it) back out of the tile loop.
#ifdef MC_EMU
__device__ inline int lane_now() { return (int)(threadIdx.x & 63); }
#else
__device__ __forceinline__ int lane_now() {
    return opaque((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
}
#endif

// what the epilogue reads (by value: the callers fetch it where they need it)
struct EpiArgs {
    half_t* C;
    const half_t* R;
    int M, N, ldc, ldr;
    float alpha;
    uint32_t bytesC, bytesR;
    bool has_bias;
    // GNS epilogues only: GroupNorm partial sums of the tile's OUTPUT for the norm that reads it next (resnet.py:197, attention.py:105,
    // motion_module.py:145): float[frame][chunk][32 groups][2] with chunks of 32 TM rows, gn_hw tokens per frame, gn_cpg channels
    // per group (N / 32)
    float* gn_partial;
    int gn_hw, gn_cpg;
};

// the wave's 160 bias values of output columns [n0, n0 + 160) of bias row `row` (row-major [rows][N] fp32): lane l < 40 gets
// columns 4 l .. 4 l + 3 (out of range: zeros) - ONE load, issued a k-stage or more ahead of the epilogue
template <int WCOL = 160>
__device__ __forceinline__ f32x4 load_bias4(const float* bias, int row, int N, int n0) {
    const GBuf bufB = make_gbuf(bias + (size_t)row * N, (uint32_t)N * 4u);
    const int ln = lane_now(), n = n0 + 4 * ln;
    return __builtin_bit_cast(f32x4, gbuf_ld8(bufB, (ln < WCOL / 4 && n < N) ? (uint32_t)n * 4u : kOOB));
}

// Epilogue of one wave tile (64 x 160 at (mw0, nw0)) through a 32 x 80-column image `stg`.  EPI 0: two column halves per
// 32-row block (four passes); EPI 1 (fused GEGLU, 80 outputs per row): two passes.
//
// vmcnt retires in issue order, so waiting for a LOAD also waits for every STORE issued before it.  The epilogue is
// therefore written so that no wait ever reaches the stores of the pass in front of it:
//   * STRAIGHT-LINE code, every load / store a buffer instruction with the hardware range check instead of a predicate:
//     the compiler's own waits are exact counts (a load inside an exec-masked branch is waited for with vmcnt(0)), and a wave
//     issues the same number of vector-memory instructions for every tile - the callers' vmcnt arithmetic depends on it;
//   * the bias (one row: rows_per_batch >= M) does not come from vector memory at all here: the caller fetched the wave's 160
//     values with ONE load a k-stage ago (`bias4`: lane l holds columns 4 l .. 4 l + 3), they go through a wave-private LDS
//     strip `bstrip` and every accumulator chunk reads its four with a broadcast ds_read_b128;
//   * the residual rows of pass p + 1 are requested BEFORE the stores of pass p are issued (into the registers pass p's
//     residual rows have just left).
// Bias added in fp32 before the fp16 rounding, residual added to the rounded value: the reference's order
// (attention.py:293-299) and gemm5's, bit for bit.
//
// GNS (round 6): the wave also leaves the GroupNorm statistics of what it stores - (sum, sum of squares) of the ROUNDED fp16
// outputs (after the residual: the values the next GroupNorm reads) per group over the wave tile's 32 TM rows, written to
// gn_partial[frame][chunk][group] where gn_block_stats of the consuming kernel adds the chunks in order: the statistics pass over
// the tensor (gn_partial_kernel: one launch and one read of the tensor per GroupNorm) is gone.  Requires that a wave tile
// lies inside one frame (hw % (32 TM) == 0, M % hw == 0) and that its 160 columns hold whole groups (160 % cpg == 0, cpg even):
// the dispatcher checks.  Two fp16 columns per v_dot2_f32_f16; lanes -> groups through the (by then idle) image; every
// (frame, chunk, group) slot has ONE writer and the additions have a fixed order: deterministic.
// TNV = column blocks of the wave tile (5: 160 columns, two 80-column passes per 32-row block; 4: 128 columns, two 64-column passes).
template <int EPI, int RES, int TM, int GNS = 0, int TNV = g5::TN>
__device__ __forceinline__ void tile_epilogue(const EpiArgs& a, f32x16 (&acc)[TNV][TM], char* stg, char* bstrip, f32x4 bias4,
                                              int mw0, int nw0) {
    static_assert(!GNS || (EPI == 0 && TNV == 5), "statistics of the plain 160-column epilogue only");
    using namespace g5;
    constexpr int H = EPI == 1 ? 1 : 2;               // passes per 32-row block
    constexpr int NP = TM * H;                        // passes
    constexpr int CPP = 4 * TNV / H;                  // accumulator chunks (i, q) per pass
    constexpr int CB = TNV == 5 ? 5 : 4;              // chunks per scheduling group
    constexpr int HC = 16 * TNV;                      // image columns: 80 / 64 (EPI 1: outputs per row)
    constexpr int PITCH = 2 * HC + 16, SEGS = HC / 8, RPI_OUT = 64 / SEGS, NIT = (32 + RPI_OUT - 1) / RPI_OUT;
    static_assert(PITCH <= RSG && CPP % CB == 0, "image fits the slot laid out for 80 columns");
    const int M = a.M, N = a.N, ldc = a.ldc, ldr = a.ldr;
    const float alpha = a.alpha;
    constexpr bool has_r = EPI == 0 && RES != 0;
    const bool has_b = a.has_bias;
    const GBuf bufC = make_gbuf(a.C, a.bytesC);
    const GBuf bufR = make_gbuf(has_r ? (const void*)a.R : (const void*)a.C, has_r ? a.bytesR : 0u);
    const int lane = lane_now();
    const int l31 = lane & 31, lhi = lane >> 5;
    const int seg = lane % SEGS, rsel = lane / SEGS;  // lanes 60..63: rsel == 6 -> out of range
    const int nout = EPI == 1 ? N / 2 : N;
    if (has_b) {
        if (lane < 8 * TNV) *reinterpret_cast<f32x4*>(bstrip + lane * 16) = bias4;
    }
    float gsum[H][4], gsq[H][4];   // GNS: column-pair sums of the lane's 8 columns per column half, over both 32-row blocks
#pragma unroll
    for (int hh = 0; hh < H; ++hh)
#pragma unroll
        for (int e = 0; e < 4; ++e) gsum[hh][e] = gsq[hh][e] = 0.f;
    half8_t rres[NIT];     // residual rows of the pass being written (requested one pass ahead, see below)
    auto load_r = [&](int pass, half8_t* dst) {
        const int mrow = mw0 + 32 * (pass / H);
        const int ncol = nw0 + HC * (pass % H) + seg * 8;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = it * RPI_OUT + rsel, m = mrow + r;
            const bool ok = rsel < RPI_OUT && r < 32 && m < M && ncol < nout;
            dst[it] = gbuf_ld8(bufR, ok ? ((uint32_t)m * (uint32_t)ldr + (uint32_t)ncol) * 2u : kOOB);
        }
    };
    if (has_r) load_r(0, rres);
    wave_lds_sync();                                  // the bias strip is readable
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        const int j = pass / H, h = pass % H;
        const int mrow = mw0 + 32 * j;
        const int ncol = EPI == 1 ? nw0 / 2 + seg * 8 : nw0 + HC * h + seg * 8;   // first output column of the lane's segment
#pragma unroll
        for (int c0 = 0; c0 < CPP; c0 += CB) {
            f32x4 bv[CB];
            if (has_b) {
#pragma unroll
                for (int c = 0; c < CB; ++c) {
                    const int iq = h * CPP + c0 + c;
                    bv[c] = *reinterpret_cast<const f32x4*>(bstrip + (32 * (iq >> 2) + 8 * (iq & 3) + 4 * lhi) * 4);
                }
            }
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                const int iq = h * CPP + c0 + c, i = iq >> 2, q = iq & 3;
                const int cl = 32 * i + 8 * q + 4 * lhi;   // column within the wave's 160
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e] * alpha;
                if (has_b) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += bv[c][e];
                }
                if (EPI == 1) {   // fused GEGLU: columns are (h, gate) pairs
                    half2_t o;
                    o[0] = to_half(v[0] * gelu_f(v[1]));
                    o[1] = to_half(v[2] * gelu_f(v[3]));
                    *reinterpret_cast<half2_t*>(stg + l31 * PITCH + cl) = o;
                } else {
                    half4_t o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = to_half(v[e]);
                    *reinterpret_cast<half4_t*>(stg + l31 * PITCH + (cl - HC * h) * 2) = o;
                }
            }
            MC_SCHED_FENCE();   // keeps hipcc from hoisting every chunk's arithmetic to the top (spills: each reload is a vmcnt(0))
        }
        wave_lds_sync();
        // image rows -> registers (+ residual); THEN the next pass's residual rows are requested (into the registers this pass
        // has just finished with) and only then this pass's stores are issued: a wait for those loads never reaches them
        half8_t o[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = it * RPI_OUT + rsel;
            o[it] = *reinterpret_cast<const half8_t*>(stg + min(r, 31) * PITCH + seg * 16);
            if (has_r) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[it][e] = to_half((float)o[it][e] + (float)rres[it][e]);
            }
        }
        MC_SCHED_FENCE();
        if (has_r && pass + 1 < NP) load_r(pass + 1, rres);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = it * RPI_OUT + rsel, m = mrow + r;
            const bool ok = rsel < RPI_OUT && r < 32 && m < M && ncol < nout;
            gbuf_st8(bufC, ok ? ((uint32_t)m * (uint32_t)ldc + (uint32_t)ncol) * 2u : kOOB, o[it]);
            if constexpr (GNS != 0) {
                half2_t one2;
                one2[0] = one2[1] = (half_t)1.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    half2_t v;
                    v[0] = ok ? o[it][2 * e] : (half_t)0.0f;
                    v[1] = ok ? o[it][2 * e + 1] : (half_t)0.0f;
                    gsum[h][e] = dot2acc(v, one2, gsum[h][e]);
                    gsq[h][e] = dot2acc(v, v, gsq[h][e]);
                }
            }
        }
        wave_lds_sync();   // the image is rewritten by the next pass
        MC_SCHED_FENCE();
    }
    if constexpr (GNS != 0) {
        // lanes -> groups: every lane leaves its 8 pair sums in the image as [rsel][pair 0 .. 79] (sum, sumsq); then 64 / G lanes
        // per group (G = 160 / cpg groups in the wave's columns) add a strided share of the group's 6 x cpg / 2 entries in a fixed
        // order and a butterfly joins them.  (The image is idle: the last pass has been read back and waited for.)
        f32x2* red = reinterpret_cast<f32x2*>(stg);
        if (rsel < RPI_OUT) {
#pragma unroll
            for (int hh = 0; hh < H; ++hh)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f32x2 pr;
                    pr[0] = gsum[hh][e];
                    pr[1] = gsq[hh][e];
                    red[rsel * 80 + 40 * hh + 4 * seg + e] = pr;
                }
        }
        wave_lds_sync();
        const int ppg = a.gn_cpg >> 1;            // column pairs per group: 5 / 10 / 20
        const int G = 80 / ppg;                   // groups in the wave's 160 columns: 16 / 8 / 4
        const int LPG = 64 / G;                   // lanes per group: 4 / 8 / 16
        const int g = lane / LPG, sub = lane % LPG;
        const int items = RPI_OUT * ppg;          // (rsel, pair) entries of one group
        float s1 = 0.f, s2 = 0.f;
        for (int it = sub; it < items; it += LPG) {
            const f32x2 v = red[(it / ppg) * 80 + g * ppg + it % ppg];
            s1 += v[0];
            s2 += v[1];
        }
        for (int mk = 1; mk < LPG; mk <<= 1) {
            s1 += shfl_xor(s1, mk);
            s2 += shfl_xor(s2, mk);
        }
        if (sub == 0 && mw0 < M) {
            const int rows = 32 * TM, frame = mw0 / a.gn_hw, chunk = (mw0 % a.gn_hw) / rows, pch = a.gn_hw / rows;
            f32x2 pr;
            pr[0] = s1;
            pr[1] = s2;
            reinterpret_cast<f32x2*>(a.gn_partial)[((size_t)frame * pch + chunk) * 32 + nw0 / a.gn_cpg + g] = pr;
        }
        wave_lds_sync();
    }
}

}  // namespace mc
