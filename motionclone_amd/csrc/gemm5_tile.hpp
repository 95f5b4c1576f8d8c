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
//   <256, 160, 4, 3>  (round 4) 4 waves as 4 x 1, wave tile 64 x 160, ring of 3 stages (78 KiB): TWO workgroups per CU that
//                     run out of step - one tile's prologue / epilogue (residual read, stores) under the other's k-loop.  Same
//                     per-wave code as the default (TM = 2), 1.44x the operand bytes per MFMA (A is fetched once per 160
//                     columns instead of once per 320).  For the short-K launches of about one wave of tiles.
template <int BM, int BN_ = BN, int NW_ = NW, int NS_ = NS>
struct Tile {
    static constexpr int BNT = BN_, NWV = NW_, NSV = NS_, NTH = NW_ * 64;
    static constexpr int WNW = BN_ / 160, WMW = NW_ / WNW;   // waves along N / along M
    static constexpr int TM = BM / (32 * WMW);
    static constexpr int RA = BM / RPI / NW_;            // activation row groups per wave and stage
    static constexpr int WB = (BN_ / RPI) / NW_;         // weight row groups every wave moves per stage ...
    static constexpr int WX = (BN_ / RPI) % NW_;         // ... and the first WX waves one more
    static constexpr int STAGE = (BM + BN_) * ROWB;      // 36864 / 28672 / 26624
    static constexpr int A_BYTES = BM * ROWB;
    static constexpr int LB = RA + WB, LA = RA + WB + 1; // LDS-DMA instructions per stage: waves >= WX / waves < WX
    static constexpr size_t SMEM = (size_t)NS_ * STAGE;
    static_assert(BN_ % 160 == 0 && NW_ % WNW == 0 && BM % (32 * WMW) == 0 && BM % (RPI * NW_) == 0, "tile / wave grid");
    static_assert(WX > 0 && NW_ * STG <= NS_ * STAGE, "epilogue image must fit the ring");
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

}  // namespace mc
