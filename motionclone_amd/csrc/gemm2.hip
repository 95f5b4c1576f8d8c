// GEMM / implicit-conv kernel, second generation (same contract as gemm.hip, see gemm_params.hpp):
//   * operand tiles go global -> LDS directly (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction); the
//     XOR swizzle of the 16-byte LDS slots is applied on the SOURCE address (the LDS image of an LDS-DMA is
//     lane-linear), conv padding and M/N tails are the buffer descriptor's out-of-range zero fill;
//   * XCD-aware tile order: block id b runs on XCD b % 8 (observed dispatch; used for speed only), so all
//     N-tiles of one M-tile are given consecutive slots of ONE XCD and re-read the activation tile from its L2;
//   * the accumulators are staged through LDS once and leave as whole 256-byte row segments (16 B per lane,
//     coalesced), with bias / residual / alpha - or the fused GEGLU of the feed-forward's first Linear
//     (diffusers FeedForward, reference attention.py:211) - applied in that pass.
#include "gemm_epilogue.hpp"

namespace mc {

// NS = LDS stages.  NS == 2: loads one K tile ahead, __syncthreads() per tile (2 workgroups per CU).
// NS == 3: loads two K tiles ahead; the wait is a COUNTED s_waitcnt vmcnt(8) (= leave the newest tile's 8
// LDS-DMA instructions of this wave in flight) followed by a raw s_barrier - a __syncthreads() here would drain
// the whole queue.  96 KiB of LDS, one workgroup per CU.
template <int MODE, int BM, int BN, int NS>
__global__ __launch_bounds__(256) void gemm2_kernel(GemmParams p, uint32_t bytesA, uint32_t bytesA2, uint32_t bytesW,
                                                     int tilesM, int tilesN) {
    constexpr int RA = BM / 32;  // 8-row groups staged per wave (A)
    constexpr int RW = BN / 32;
    constexpr int TM = BM / 64;
    constexpr int TN = BN / 64;
    constexpr int CS = BN + 4;   // fp32 row stride of the epilogue staging tile
    MC_DYN_SMEM(smem);
    char* sA = smem;                  // [NS][BM][128 B]
    char* sW = smem + NS * BM * 128;  // [NS][BN][128 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // ---- XCD-aware tile assignment ----
    const int pid = blockIdx.x;
    const int xcd = pid & 7, local = pid >> 3;
    const int tn = local % tilesN;
    const int tm = (local / tilesN) * 8 + xcd;
    if (tm >= tilesM) return;  // whole workgroup
    const int m0 = tm * BM, n0 = tn * BN;

    const GBuf bufA = make_gbuf(p.A, bytesA);
    const GBuf bufA2 = make_gbuf(p.A2 ? p.A2 : p.A, p.A2 ? bytesA2 : bytesA);
    const GBuf bufW = make_gbuf(p.W, bytesW);

    // lane -> (row within its 8-row group, physical 16-byte slot); logical slot = physical ^ swizzle(row)
    const int rsub = lane >> 3;
    const int sw = (((wave & 1) << 2) | (lane >> 4)) & 7;  // == ((row >> 1) & 7) for every group of this wave
    const int lslot = (lane & 7) ^ sw;

    int a_valid[RA], a_pix[RA], a_oy[RA], a_ox[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        int m = m0 + (wave + 4 * i) * 8 + rsub;
        a_valid[i] = m < p.M;
        if (MODE == DENSE) {
            a_pix[i] = m;
            a_oy[i] = a_ox[i] = 0;
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
    uint32_t w_off[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        int n = n0 + (wave + 4 * i) * 8 + rsub;
        w_off[i] = n < p.N ? (uint32_t)n * (uint32_t)p.K * 2u + (uint32_t)lslot * 16u : kOOB;
    }

    auto issue_tiles = [&](int kt, int buf) {
        const int k0 = kt * BK;
        int tap = 0, c0 = k0;
        if (MODE != DENSE) {  // K order: 64-channel tile major, tap minor (the 9 taps of a channel tile are adjacent)
            const int ct = kt / 9;
            tap = kt - 9 * ct;
            c0 = ct * BK;
        }
        const bool second = c0 >= p.c1;
        const int ld = second ? p.lda2 : p.lda;
        const int cc = (second ? c0 - p.c1 : c0) + lslot * 8;
        const int ky = tap / 3, kx = tap - 3 * (tap / 3);
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            bool ok = a_valid[i];
            int row;
            if (MODE == DENSE) {
                row = a_pix[i];
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
                    int uy = a_oy[i] + ky - 1, ux = a_ox[i] + kx - 1;
                    ok = ok && uy >= 0 && uy < p.Ho && ux >= 0 && ux < p.Wo;
                    iy = uy >> 1;
                    ix = ux >> 1;
                } else {
                    int ty = a_oy[i] + 1 - ky, tx = a_ox[i] + 1 - kx;
                    ok = ok && ty >= 0 && tx >= 0 && !(ty & 1) && !(tx & 1);
                    iy = ty >> 1;
                    ix = tx >> 1;
                    ok = ok && iy < p.Hs && ix < p.Ws;
                }
                row = a_pix[i] + iy * p.Ws + ix;
            }
            uint32_t voff = ok ? ((uint32_t)row * (uint32_t)ld + (uint32_t)cc) * 2u : kOOB;
            char* dst = sA + buf * BM * 128 + (wave + 4 * i) * 1024;
            if (second)
                glds16(bufA2, voff, dst);
            else
                glds16(bufA, voff, dst);
        }
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            uint32_t voff = w_off[i] == kOOB ? kOOB : w_off[i] + (uint32_t)k0 * 2u;
            glds16(bufW, voff, sW + buf * BN * 128 + (wave + 4 * i) * 1024);
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
    issue_tiles(0, 0);
    if (NS == 3) {
        if (nk > 1) issue_tiles(1, 1);
    } else {
        __syncthreads();
    }
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (NS == 3) {
            // tile kt must have landed (own loads: counted wait; everybody's: barrier).  The barrier also means
            // every wave is done reading stage (kt-1)%3, which the loads issued next overwrite.
            if (kt + 1 < nk)
                wait_vmcnt_le<RA + RW>();
            else
                wait_vmcnt_le<0>();
            raw_barrier();
            if (kt + 2 < nk) issue_tiles(kt + 2, buf >= 1 ? buf - 1 : 2);
        } else {
            if (kt + 1 < nk) issue_tiles(kt + 1, buf ^ 1);
        }
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
        if (NS == 3) {
            buf = buf == 2 ? 0 : buf + 1;
        } else {
            __syncthreads();
            buf ^= 1;
        }
    }
    if (NS == 3) __syncthreads();  // all fragment reads done before the staging tile overwrites the operands

    // ---- epilogue: accumulators -> LDS (fp32, padded rows) -> coalesced row segments ----
    float* Cs = reinterpret_cast<float*>(smem);
    constexpr int TPR = BN / 8;        // threads per row, 8 columns each
    constexpr int RPP = 256 / TPR;     // rows per store iteration
    constexpr int NIT = BM / RPP;
    Epilogue<TPR, RPP, NIT, CS> ep;
    ep.init(p, tid, n0, m0, min(m0 + BM, p.M) - 1);
    ep.prefetch(p, m0, BM);            // residual rows in flight across the staging + barrier
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e] * p.alpha;
                *reinterpret_cast<f32x4*>(Cs + (wm0 + 32 * j + l31) * CS + wn0 + 32 * i + 8 * q + 4 * lhi) = v;
            }
    __syncthreads();
    ep.store(p, Cs, m0, BM);
}

template <int MODE>
static int launch2_mode(const GemmParams& p, uint32_t bA, uint32_t bA2, uint32_t bW, int small_tile, int deep,
                        hipStream_t stream) {
    if (small_tile) {
        int tM = (p.M + 63) / 64, tN = (p.N + 63) / 64;
        dim3 grid((unsigned)(((tM + 7) / 8) * 8 * tN));
        if (deep) {
            size_t smem = 3 * (64 + 64) * 128;
            MC_LAUNCH((gemm2_kernel<MODE, 64, 64, 3>), grid, dim3(256), smem, stream, p, bA, bA2, bW, tM, tN);
        } else {
            size_t smem = 2 * (64 + 64) * 128;
            MC_LAUNCH((gemm2_kernel<MODE, 64, 64, 2>), grid, dim3(256), smem, stream, p, bA, bA2, bW, tM, tN);
        }
    } else {
        int tM = (p.M + 127) / 128, tN = (p.N + 127) / 128;
        dim3 grid((unsigned)(((tM + 7) / 8) * 8 * tN));
        if (deep) {
            size_t smem = 3 * (128 + 128) * 128;
            allow_big_smem(gemm2_kernel<MODE, 128, 128, 3>, smem);
            MC_LAUNCH((gemm2_kernel<MODE, 128, 128, 3>), grid, dim3(256), smem, stream, p, bA, bA2, bW, tM, tN);
        } else {
            size_t smem = 128 * (128 + 4) * 4;  // epilogue staging tile (> the 64 KiB of operand buffers)
            allow_big_smem(gemm2_kernel<MODE, 128, 128, 2>, smem);
            MC_LAUNCH((gemm2_kernel<MODE, 128, 128, 2>), grid, dim3(256), smem, stream, p, bA, bA2, bW, tM, tN);
        }
    }
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

// returns MC_ERR_UNSUPPORTED when an operand does not fit a 2 GiB buffer descriptor (caller falls back to v1)
int gemm2_dispatch(const GemmParams& p, int mode, int small_tile, int deep, size_t rowsA, hipStream_t stream) {
    size_t bytesA = (rowsA * (size_t)p.lda) * 2, bytesA2 = p.A2 ? (rowsA * (size_t)p.lda2) * 2 : 0;
    size_t bytesW = (size_t)p.N * p.K * 2;
    const size_t lim = 0x7FFFFFF0u;
    if (bytesA > lim || bytesA2 > lim || bytesW > lim) return MC_ERR_UNSUPPORTED;
    switch (mode) {
        case DENSE: return launch2_mode<DENSE>(p, bytesA, bytesA2, bytesW, small_tile, deep, stream);
        case CONV_S1: return launch2_mode<CONV_S1>(p, bytesA, bytesA2, bytesW, small_tile, deep, stream);
        case CONV_S2: return launch2_mode<CONV_S2>(p, bytesA, bytesA2, bytesW, small_tile, deep, stream);
        case CONV_UP: return launch2_mode<CONV_UP>(p, bytesA, bytesA2, bytesW, small_tile, deep, stream);
        default: return launch2_mode<TCONV_S2>(p, bytesA, bytesA2, bytesW, small_tile, deep, stream);
    }
}

}  // namespace mc
