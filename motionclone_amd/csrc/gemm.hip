// MFMA GEMM family for the UNet3D hot path:
//   C[M,N] = alpha * (A[M,K] . W[N,K]^T) + bias[batch(m)][n] + R[m][n]
// with the A operand produced on the fly by one of five "gather" modes so that
// nn.Linear, 1x1 conv, 3x3 conv (stride 1 / stride 2 / fused nearest-2x upsample),
// their data-gradients, and the skip-connection concat all run through ONE kernel
// on channels-last [frames, H, W, C] fp16 activations:
//   DENSE   : A row m = token m, optional second source for the channel concat
//   CONV_S1 : 3x3, stride 1, pad 1           (reference resnet.py:148,168; unet.py:98,249)
//   CONV_S2 : 3x3, stride 2, pad 1           (reference resnet.py:94)
//   CONV_UP : 3x3 on a nearest-2x upsampled source, never materialised (resnet.py:65,78)
//   TCONV_S2: data-gradient of CONV_S2 (scatter written as a gather)
// K is ordered tap-major / channel-minor (k = tap*Ctot + c), so a 64-wide K tile
// is one contiguous 128-byte run of channels of one (shifted) pixel.
//
// Tiling (gfx950): 256 threads = 4 waves as 2x2; block tile BMxBN in {128x128, 64x64},
// BK = 64; v_mfma_f32_32x32x16_f16 with the weight tile as the MFMA "A" operand so
// that each lane ends up with 4 consecutive output channels of one token (8-byte
// stores, 8-byte bias/residual loads).  Operand tiles are staged global -> VGPR ->
// LDS (16 B per lane, XOR-swizzled 16-byte slots: conflict-free ds_read_b128),
// double-buffered, one barrier per K tile, next tile's global loads issued before
// the current tile's MFMAs.
#include "gemm_params.hpp"

namespace mc {


template <int MODE, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
    constexpr int RA = BM / 32;  // A rows staged per thread
    constexpr int RW = BN / 32;
    constexpr int TM = BM / 64;  // 32x32 MFMA tiles per wave along M
    constexpr int TN = BN / 64;
    MC_DYN_SMEM(smem);
    char* sA = smem;                       // [2][BM][128 B]
    char* sW = smem + 2 * BM * 128;        // [2][BN][128 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int vcol = tid & 7;
    const int rbase = tid >> 3;  // 0..31

    // ---- per-thread row descriptors (fixed over the K loop) ----
    int a_valid[RA];
    int a_pix[RA];  // DENSE: row index m; conv: frame * Hs * Ws
    int a_oy[RA], a_ox[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        int m = m0 + rbase + 32 * i;
        a_valid[i] = m < p.M;
        if (MODE == DENSE) {
            a_pix[i] = m;
            a_oy[i] = 0;
            a_ox[i] = 0;
        } else {
            int hw = p.Ho * p.Wo;
            int fr = m / hw;
            int rem = m - fr * hw;
            int oy = rem / p.Wo;
            a_pix[i] = fr * p.Hs * p.Ws;
            a_oy[i] = oy;
            a_ox[i] = rem - oy * p.Wo;
        }
    }
    const half_t* w_ptr[RW];
    int w_valid[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        int n = n0 + rbase + 32 * i;
        w_valid[i] = n < p.N;
        w_ptr[i] = p.W + (size_t)(w_valid[i] ? n : 0) * p.K + vcol * 8;
    }

    half8_t ra[RA], rw[RW];

    auto load_tiles = [&](int kt) {
        const int k0 = kt * BK;
        int tap = 0, c0 = k0;
        if (MODE != DENSE) {  // K order: 64-channel tile major, tap minor (the 9 taps of a channel tile are adjacent)
            const int ct = kt / 9;
            tap = kt - 9 * ct;
            c0 = ct * BK;
        }
        const half_t* src = p.A;
        int ld = p.lda;
        int cc = c0;
        if (c0 >= p.c1) {
            src = p.A2;
            ld = p.lda2;
            cc = c0 - p.c1;
        }
        const int ky = tap / 3, kx = tap - 3 * (tap / 3);
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            bool ok = a_valid[i];
            size_t row;
            if (MODE == DENSE) {
                row = (size_t)a_pix[i];
            } else {
                int iy, ix;
                if (MODE == CONV_S1) {
                    iy = a_oy[i] + ky - 1;
                    ix = a_ox[i] + kx - 1;
                    ok = ok && iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws;
                } else if (MODE == CONV_S2) {
                    iy = 2 * a_oy[i] + ky - p.s2_pad;
                    ix = 2 * a_ox[i] + kx - p.s2_pad;
                    ok = ok && iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws;
                } else if (MODE == CONV_UP) {
                    int uy = a_oy[i] + ky - 1, ux = a_ox[i] + kx - 1;  // on the upsampled grid
                    ok = ok && uy >= 0 && uy < p.Ho && ux >= 0 && ux < p.Wo;
                    iy = uy >> 1;
                    ix = ux >> 1;
                } else {  // TCONV_S2: y = 2*sy + ky - 1  <=>  sy = (y + 1 - ky) / 2
                    int ty = a_oy[i] + 1 - ky, tx = a_ox[i] + 1 - kx;
                    ok = ok && ty >= 0 && tx >= 0 && !(ty & 1) && !(tx & 1);
                    iy = ty >> 1;
                    ix = tx >> 1;
                    ok = ok && iy < p.Hs && ix < p.Ws;
                }
                row = (size_t)(a_pix[i] + iy * p.Ws + ix);
            }
            ra[i] = ok ? ld8(src + row * ld + cc + vcol * 8) : zero8();
        }
#pragma unroll
        for (int i = 0; i < RW; ++i) rw[i] = w_valid[i] ? ld8(w_ptr[i] + k0) : zero8();
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            int r = rbase + 32 * i;
            *reinterpret_cast<half8_t*>(sA + buf * BM * 128 + lds_off(r, vcol)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            int r = rbase + 32 * i;
            *reinterpret_cast<half8_t*>(sW + buf * BN * 128 + lds_off(r, vcol)) = rw[i];
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm0 = (wave & 1) * (BM / 2);
    const int wn0 = (wave >> 1) * (BN / 2);
    const int l31 = lane & 31, lhi = lane >> 5;

    const int nk = p.K / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);
        const char* bA = sA + buf * BM * 128;
        const char* bW = sW + buf * BN * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            half8_t fa[TM], fw[TN];
#pragma unroll
            for (int j = 0; j < TM; ++j)
                fa[j] = *reinterpret_cast<const half8_t*>(bA + lds_off(wm0 + 32 * j + l31, 2 * ks + lhi));
#pragma unroll
            for (int i = 0; i < TN; ++i)
                fw[i] = *reinterpret_cast<const half8_t*>(bW + lds_off(wn0 + 32 * i + l31, 2 * ks + lhi));
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = mfma32(fw[i], fa[j], acc[i][j]);
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds D[n = nb + (r&3) + 8*(r>>2) + 4*lhi][m = mb + l31] ----
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + wm0 + 32 * j + l31;
        if (m >= p.M) continue;
        const float* brow = p.bias ? p.bias + (size_t)(m / p.rows_per_batch) * p.N : nullptr;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn0 + 32 * i + 8 * q + 4 * lhi;
                if (n >= p.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e] * p.alpha;
                if (brow) {
                    f32x4 b = *reinterpret_cast<const f32x4*>(brow + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += b[e];
                }
                if (p.R) {
                    half4_t r = ld4(p.R + (size_t)m * p.ldr + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
                }
                half4_t o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = to_half(v[e]);
                st4(p.C + (size_t)m * p.ldc + n, o);
            }
        }
    }
}

template <int MODE>
static int launch_mode(const GemmParams& p, int small_tile, hipStream_t stream) {
    if (small_tile) {
        dim3 grid((p.N + 63) / 64, (p.M + 63) / 64);
        MC_LAUNCH((gemm_kernel<MODE, 64, 64>), grid, dim3(256), 2 * (64 + 64) * 128, stream, p);
    } else {
        dim3 grid((p.N + 127) / 128, (p.M + 127) / 128);
        MC_LAUNCH((gemm_kernel<MODE, 128, 128>), grid, dim3(256), 2 * (128 + 128) * 128, stream, p);
    }
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

int gemm2_dispatch(const GemmParams& p, int mode, int small_tile, int deep, size_t rowsA, hipStream_t stream);
int gemm3_dispatch(const GemmParams& p, int mode, int cfg, size_t rowsA, hipStream_t stream);  // gemm3.hip  // gemm2.hip

}  // namespace mc

using namespace mc;

extern "C" int mc_gemm_f16(const void* A, const void* A2, const void* W, void* C, const void* R,
                           const float* bias, int M, int N, int K, int lda, int lda2, int ldc, int ldr,
                           int c1, int ctot, int mode, int Hs, int Ws, int Ho, int Wo,
                           int rows_per_batch, float alpha, int flags, void* stream) {
    const int tile = flags & 0xFF;        // 0 = auto, 64, 128
    const int force_v1 = flags & 0x100;   // first-generation kernel (register-staged operands)
    const int epi = (flags & 0x200) ? 1 : 0;
    const int deep = (flags & 0x400) ? 1 : 0;
    int big_cfg = (flags >> 12) & 0xF;   // large-tile kernel geometry (gemm3.hip), 0 = choose automatically  // v2 with a 3-stage LDS ring (loads two K tiles ahead)  // fused GEGLU epilogue (weights row-interleaved h/gate)
    if (M <= 0 || N <= 0 || K <= 0) return MC_ERR_SHAPE;
    if (epi && (R || N % 8 || force_v1)) return MC_ERR_UNSUPPORTED;
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
    static const int dbg_env = getenv("MC_GEMM_DEBUG") ? atoi(getenv("MC_GEMM_DEBUG")) : 0;
    p.dbg = dbg_env;
    p.ws = nullptr;
    p.splits = 1;
    p.s2_pad = (flags & 0x800) ? 0 : 1;
    int small_tile = tile == 64;
    if (tile == 0) {
        // heuristic: fall to 64x64 tiles when 128x128 would leave most of the 256 CUs idle
        long big = (long)((M + 127) / 128) * ((N + 127) / 128);
        small_tile = big < 256;
    }
    hipStream_t s = (hipStream_t)stream;
    if (!big_cfg && !tile && !force_v1 && !deep && N % 320 == 0) {
        // measured on MI355X (profiles/r01_kernel_microbench_v3.jsonl): the 256x320 / 128x320 geometries win
        // once they still fill the 256 CUs; smaller problems stay on the 128x128 / 64x64 tiles
        long b1 = (long)((M + 255) / 256) * (N / 320), b4 = (long)((M + 127) / 128) * (N / 320);
        if (b1 >= 224) big_cfg = 1;
        else if (b4 >= 192) big_cfg = 4;
    }
    if (big_cfg) {
        size_t rowsA3 = mode == DENSE ? (size_t)M : (size_t)(M / (Ho * Wo)) * Hs * Ws;
        int rc3 = gemm3_dispatch(p, mode, big_cfg, rowsA3, s);
        if (rc3 != MC_ERR_UNSUPPORTED) return rc3;
    }
    if (!force_v1) {
        size_t rowsA = mode == DENSE ? (size_t)M : (size_t)(M / (Ho * Wo)) * Hs * Ws;
        int rc = gemm2_dispatch(p, mode, small_tile, deep, rowsA, s);
        if (rc != MC_ERR_UNSUPPORTED) return rc;
        if (epi) return MC_ERR_UNSUPPORTED;
    }
    switch (mode) {
        case DENSE: return launch_mode<DENSE>(p, small_tile, s);
        case CONV_S1: return launch_mode<CONV_S1>(p, small_tile, s);
        case CONV_S2: return launch_mode<CONV_S2>(p, small_tile, s);
        case CONV_UP: return launch_mode<CONV_UP>(p, small_tile, s);
        default: return launch_mode<TCONV_S2>(p, small_tile, s);
    }
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
    (void)mode;
    if (N % 320 || K < 2304 || M <= 0) return 1;
    const int nk = K / BK;
    long t1 = (long)((M + 255) / 256) * (N / 320), t4 = (long)((M + 127) / 128) * (N / 320);
    if (t1 >= 224) return 1;                       // 256x320 tiles already fill the chip
    int s, cfg;
    if (t1 >= 64) {                                // 64..223 big tiles: keep the efficient geometry, 2-4 K ranges
        s = (int)(256 / t1);
        cfg = 1;
    } else {                                       // fewer: 128x320 tiles, up to 8 K ranges
        if (t4 >= 256) return 1;
        s = (int)(256 / t4);
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
    if (cfg > 6) return MC_ERR_UNSUPPORTED;   // two-stage geometries only
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
    int rc = gemm3_dispatch(p, mode, cfg, rowsA, s);
    if (rc != MC_OK) return rc;
    long nthr = (long)M * (N / 4);
    MC_LAUNCH(splitk_reduce_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, ws, splits, (half_t*)C, ldc,
              (const half_t*)R, ldr, bias, M, N, rows_per_batch);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}
