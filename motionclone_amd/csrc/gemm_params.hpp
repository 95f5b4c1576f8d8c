// Shared parameter block of the GEMM / implicit-conv kernels (gemm.hip, gemm2.hip).
#pragma once
#include "mc_common.hpp"

namespace mc {

enum GemmMode { DENSE = 0, CONV_S1 = 1, CONV_S2 = 2, CONV_UP = 3, TCONV_S2 = 4 };

struct GemmParams {
    const half_t* A;
    const half_t* A2;
    const half_t* W;
    half_t* C;
    const half_t* R;
    const float* bias;
    int M, N, K;
    int lda, lda2, ldc, ldr;
    int c1;     // channels taken from A; the rest (ctot - c1) come from A2
    int ctot;   // channels per tap (conv) or K (dense).  Conv K index: k = (c / 64) * 576 + tap * 64 + (c % 64), i.e.
                // the 9 taps of one 64-channel tile are consecutive K tiles: a workgroup re-reads the same 32 KiB of
                // activations 9 times back to back (L2 hits) instead of once per pass over all channels.
    int Hs, Ws; // source grid (per frame)
    int Ho, Wo; // output grid (per frame)
    int rows_per_batch;
    float alpha;
    int epi;    // 0: bias/residual epilogue, 1: fused GEGLU (v2 kernel only)
    int s2_pad; // CONV_S2: zero rows / columns in front of the image: 1 (symmetric padding 1, the UNet) or 0 (pad only
                // right / bottom: diffusers Downsample2D(padding=0) = F.pad(x, (0, 1, 0, 1)) + stride-2 conv, the VAE encoder)
    float* ws;  // split-K: fp32 partial sums [splits][M][N] (then C/R/bias are applied by the reduce kernel); else null
    int splits; // number of K ranges (grid.y); 1 without split-K
    int dbg;    // timing experiments only (MC_GEMM_DEBUG): 1 = no global stores, 2 = no k-loop, 4 = no epilogue
    // mc_gemm_gnstats_f16 (gemm5 / gemm6 one-pass epilogues): GroupNorm partial sums of the OUTPUT, see gemm5_tile.hpp tile_epilogue
    float* gn_partial = nullptr;
    int gn_hw = 0;
};

// normalisation applied to the rows of A inside the K = 320 streaming kernel (gemm4.hip, mc_norm_gemm_f16)
struct G4Norm {
    int kind;               // 0 none, 1 LayerNorm, 2 per-frame affine
    const float* gamma;
    const float* beta;
    const float* pe;        // kind 1: [nframes_pe][K] or null
    int hw, nframes_pe;
    float eps;
    float* stats;           // kind 1: out [M][2] (mean, rstd) or null;  kind 2: out [frames][32][2]
    const float* partial;   // kind 2: per-chunk (sum, sumsq) of gn_partial_kernel
    int nchunk;
    float gn_n;             // kind 2: elements per group (hw * channels per group)
};

constexpr int BK = 64;

__device__ __forceinline__ int lds_off(int row, int v) {
    // byte offset of 16-byte slot v (0..7) of a 128-byte row; slots XOR-swizzled by (row>>1)&7
    return row * 128 + ((v ^ ((row >> 1) & 7)) << 4);
}

}  // namespace mc
