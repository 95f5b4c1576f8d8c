// GEMM / implicit-conv kernel, large-tile generation (contract: gemm_params.hpp, same as gemm.hip / gemm2.hip).
//
// Why: on gfx950 every LDS-DMA instruction (buffer_load_dwordx4 ... lds, 1 KiB) costs on the order of 100 issue
// cycles, so the staging cost per workgroup tile must be amortised over far more MFMA work than a 128x128 tile
// with four waves offers.  Here a workgroup is 8 waves (NWM x NWN), each owning a (32*TM) x (32*TN) block of
// v_mfma_f32_32x32x16_f16 tiles; the block tile is BM x BN with BN a multiple of 320 where the layer widths of the
// SD-1.5 UNet (320, 640, 960, 1280, 1920, 2560, ...) allow it, so no N padding is computed.
// Everything else is as in gemm2.hip: operand tiles global -> LDS directly with the XOR swizzle on the source
// address, hardware zero fill for conv padding / tails, XCD-aware tile order, accumulators leave through an LDS
// staging tile as coalesced 16-byte row segments with the bias / residual / alpha / fused-GEGLU epilogue.
#include "gemm_epilogue.hpp"

namespace mc {

// byte offset of 16-byte slot v of a row of BKT halfs; slots XOR-swizzled so that every 16-lane group of a
// ds_read_b128 (rows l31, fixed logical slot) covers all 64 banks: 128-byte rows by (row>>1)&7, 64-byte rows by
// (row>>2)&3 (rows {r, 12+r, 20+r, 24+r} of a lane group then take the four distinct slots of their bank quarter)
template <int BKT>
__device__ __forceinline__ int lds_off_t(int row, int v) {
    if (BKT == 64) return row * 128 + ((v ^ ((row >> 1) & 7)) << 4);
    return row * 64 + ((v ^ ((row >> 2) & 3)) << 4);
}

// BKT: K depth of one staged tile (64, or 32 to halve the operand LDS so that two workgroups share a CU and one's
//      epilogue stores overlap the other's MFMAs - the K = 320 GEMMs spend half their time in the epilogue).
// PJ:  32-row accumulator sub-tiles written per epilogue pass (TM = all of a wave-row at once, 1 = small staging tile).
// WPE: minimum waves per SIMD the register allocation must allow.
// NS:  LDS stages.  2 = load tile k+1 while computing k (one __syncthreads per tile); >= 3 = ring with NS-1 tiles in
//      flight, counted vmcnt waits and one raw barrier per tile: a k-step of the 2-stage loop measures ~2.3 us
//      against ~1.1 us of MFMA work because each step exposes the full LDS-DMA landing latency.
template <int MODE, int BM, int BN, int NWM, int NWN, int BKT, int PJ, int WPE, int NS>
__global__ __launch_bounds__(64 * NWM * NWN, WPE) void gemm3_kernel(GemmParams p, uint32_t bytesA, uint32_t bytesA2,
                                                                     uint32_t bytesW, int tilesM, int tilesN) {
    constexpr int NW = NWM * NWN;
    constexpr int NT = 64 * NW;
    constexpr int TM = BM / NWM / 32;
    constexpr int TN = BN / NWN / 32;
    constexpr int ROWB = BKT * 2;        // bytes per staged row
    constexpr int SPR = ROWB / 16;       // 16-byte slots per row
    constexpr int RPI = 64 / SPR;        // rows moved by one wave-wide LDS-DMA instruction (1 KiB)
    constexpr int RA = BM / RPI / NW;    // row groups staged per wave
    constexpr int RWLO = BN / RPI / NW;  // weight row groups per wave; the first EXTRA waves take one more
    constexpr int EXTRA = (BN / RPI) % NW;
    constexpr int RW = RWLO + (EXTRA ? 1 : 0);
    constexpr int WROWS = 32 * PJ;       // rows of one epilogue pass
    constexpr int CS = BN + 4;
    static_assert(BM % (32 * NWM) == 0 && BN % (32 * NWN) == 0 && BM % (RPI * NW) == 0 && BN % RPI == 0, "tile");
    static_assert(NS >= 2 && (EXTRA == 0 || NS > 2), "uneven weight staging needs the counted-wait ring");
    static_assert(NW % 2 == 0, "swizzle constant assumes an even wave count");
    static_assert(TM % PJ == 0 && (BKT == 64 || BKT == 32), "epilogue pass / K depth");
    MC_DYN_SMEM(smem);
    char* sA = smem;                    // [NS][BM][ROWB]
    char* sW = smem + NS * BM * ROWB;   // [NS][BN][ROWB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int pid = blockIdx.x;
    const int xcd = pid & 7, local = pid >> 3;
    const int tn = local % tilesN;
    // Linear layers: the N-tiles of one M-tile sit on one XCD (consecutive slots) and share the activation tile in its L2.
    // Convolutions: every XCD owns a CONTIGUOUS range of M-tiles instead, so that vertically adjacent tiles - which
    // read each other's halo image rows through the shifted taps - run on the same XCD at about the same time and
    // meet in its L2 (256-row tiles = 4 image rows at 64 px: 6 rows are read per tile, 1.5x if the halo is private).
    // Measured (tools/conv_bench.py, tools/pmc_ab.sh): the level-0 conv (one N-tile) 282 -> 269 us and 239 -> 200 MB of HBM
    // traffic per launch (1.41x -> 1.18x algorithmic); with two or more N-tiles the same order was 3 % slower, so those keep
    // the Linear order.
    int tm;
    if (MODE == DENSE || tilesN > 1) {
        tm = (local / tilesN) * 8 + xcd;
    } else {
        const int per = (tilesM + 7) >> 3;
        tm = (local / tilesN) < per ? xcd * per + local / tilesN : tilesM;
    }
    if (tm >= tilesM) return;
    const int m0 = tm * BM, n0 = tn * BN;

    const GBuf bufA = make_gbuf(p.A, bytesA);
    const GBuf bufA2 = make_gbuf(p.A2 ? p.A2 : p.A, p.A2 ? bytesA2 : bytesA);
    const GBuf bufW = make_gbuf(p.W, bytesW);

    // lane -> (row within the instruction's row group, physical slot); the logical slot it must fetch undoes the swizzle
    const int rsub = lane / SPR;
    const int lslot = BKT == 64 ? ((lane & 7) ^ ((((wave & 1) << 2) | (lane >> 4)) & 7)) : ((lane & 3) ^ (lane >> 4));

    int a_valid[RA], a_pix[RA], a_oy[RA], a_ox[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        int m = m0 + (wave + NW * i) * RPI + rsub;
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
    const bool w_hi = EXTRA == 0 || wave < EXTRA;   // this wave stages RW (not RW-1) weight row groups
    uint32_t w_off[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        int n = n0 + (wave + NW * i) * RPI + rsub;
        w_off[i] = n < p.N ? (uint32_t)n * (uint32_t)p.K * 2u + (uint32_t)lslot * 16u : kOOB;
    }

    auto issue_tiles = [&](int kt, int buf) {
        const int k0 = kt * BKT;
        int tap = 0, c0 = k0;
        if (MODE != DENSE) {  // K order: 64-channel tile major, tap minor (the 9 taps of a channel tile are adjacent)
            const int kt64 = BKT == 64 ? kt : kt >> 1;
            const int ct = kt64 / 9;
            tap = kt64 - 9 * ct;
            c0 = ct * 64 + (BKT == 64 ? 0 : (kt & 1) * 32);
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
            char* dst = sA + buf * BM * ROWB + (wave + NW * i) * 1024;
            if (second)
                glds16(bufA2, voff, dst);
            else
                glds16(bufA, voff, dst);
        }
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            if (i == RW - 1 && !w_hi) break;   // wave-uniform
            uint32_t voff = w_off[i] == kOOB ? kOOB : w_off[i] + (uint32_t)k0 * 2u;
            glds16(bufW, voff, sW + buf * BN * ROWB + (wave + NW * i) * 1024);
        }
    };
    // wait until at most `tiles` of this wave's staged tiles are still in flight (vmcnt retires in order)
    auto wait_tiles = [&](int tiles) {
        constexpr int LHI = RA + RW, LLO = RA + RW - 1;
        if (tiles <= 0) {
            wait_vmcnt_le<0>();
        } else if (tiles == 1) {
            if (w_hi) wait_vmcnt_le<LHI>(); else wait_vmcnt_le<LLO>();
        } else if (tiles == 2) {
            if (w_hi) wait_vmcnt_le<2 * LHI>(); else wait_vmcnt_le<2 * LLO>();
        } else {
            if (w_hi) wait_vmcnt_le<3 * LHI>(); else wait_vmcnt_le<3 * LLO>();
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wr = wave % NWM, wc = wave / NWM;
    const int wm0 = wr * (32 * TM);
    const int wn0 = wc * (32 * TN);
    const int l31 = lane & 31, lhi = lane >> 5;

    // split-K: grid.y K ranges of (almost) equal length; the kernel is otherwise unchanged, its epilogue stores raw
    // fp32 partial sums and a reduce kernel applies bias / residual (small-M, deep-K convs of the 8x8 / 16x16 levels)
    const int split = blockIdx.y;
    const int nk_all = p.K / BKT;
    const int kt_begin = (int)((long)split * nk_all / p.splits);
    const int nk = (p.dbg & 2) ? 0 : (int)((long)(split + 1) * nk_all / p.splits);
    if (p.dbg & 4) return;
    if (NS == 2) {
        if (nk > kt_begin) issue_tiles(kt_begin, 0);
        __syncthreads();
    } else {
#pragma unroll
        for (int s0 = 0; s0 < NS - 1; ++s0)
            if (kt_begin + s0 < nk) issue_tiles(kt_begin + s0, s0);
    }
    int buf = 0;
    for (int kt = kt_begin; kt < nk; ++kt) {
        if (NS == 2) {
            if (kt + 1 < nk) issue_tiles(kt + 1, buf ^ 1);
        } else {
            // tile kt landed (own loads: counted wait; everybody's: barrier).  Past the barrier every wave is also done
            // reading stage (kt-1) % NS, which the tile issued next overwrites.
            wait_tiles(min(NS - 2, nk - 1 - kt));
            raw_barrier();
            if (kt + NS - 1 < nk) issue_tiles(kt + NS - 1, buf == 0 ? NS - 1 : buf - 1);
        }
        const char* bA = sA + buf * BM * ROWB;
        const char* bW = sW + buf * BN * ROWB;
        // fragments of k-slice ks+1 are requested before the MFMAs of slice ks are issued (two register sets), so the
        // ~100+ cycle ds_read latency runs under 10 MFMAs instead of draining the matrix pipe at every wait
        half8_t fa[2][TM], fw[2][TN];
        auto load_frags = [&](int ks, half8_t* a, half8_t* w) {
#pragma unroll
            for (int j = 0; j < TM; ++j)
                a[j] = *reinterpret_cast<const half8_t*>(bA + lds_off_t<BKT>(wm0 + 32 * j + l31, 2 * ks + lhi));
#pragma unroll
            for (int i = 0; i < TN; ++i)
                w[i] = *reinterpret_cast<const half8_t*>(bW + lds_off_t<BKT>(wn0 + 32 * i + l31, 2 * ks + lhi));
        };
        load_frags(0, fa[0], fw[0]);
#pragma unroll
        for (int ks = 0; ks < BKT / 16; ++ks) {
            if (ks + 1 < BKT / 16) load_frags(ks + 1, fa[(ks + 1) & 1], fw[(ks + 1) & 1]);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = mfma32(fw[ks & 1][i], fa[ks & 1][j], acc[i][j]);
#ifndef MC_EMU
            // pin the interleave: one fragment read of the next slice behind each of the first TM+TN MFMAs
            if (ks + 1 < BKT / 16) {
#pragma unroll
                for (int r = 0; r < TM + TN; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                }
                __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - TM - TN, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
            }
#endif
        }
        if (NS == 2) {
            __syncthreads();
            buf ^= 1;
        } else {
            buf = buf == NS - 1 ? 0 : buf + 1;
        }
    }
    if (NS > 2) __syncthreads();  // all fragment reads done before the staging tile overwrites the operands

    // ---- epilogue: WROWS rows x BN columns at a time through an fp32 LDS staging tile ----
    float* Cs = reinterpret_cast<float*>(smem);
    constexpr int TPR = BN / 8;
    constexpr int RPP = NT / TPR;       // rows per store iteration (threads beyond RPP*TPR idle in the store phase)
    constexpr int NIT = (WROWS + RPP - 1) / RPP;
    constexpr int NPASS = BM / WROWS;
    constexpr int PPW = TM / PJ;        // passes per wave-row
    Epilogue<TPR, RPP, NIT, CS> ep;
    ep.init(p, tid, n0, m0, min(m0 + BM, p.M) - 1);
#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
        ep.prefetch(p, m0 + pass * WROWS, WROWS);   // residual rows in flight across the staging + barrier
        if (wr == pass / PPW) {
            const int jb = (pass % PPW) * PJ;
#pragma unroll
            for (int jj = 0; jj < PJ; ++jj)
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
                        if (PJ == TM) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[i][jj][4 * q + e] * p.alpha;
                        } else {   // runtime sub-tile index: select without dynamic register indexing
#pragma unroll
                            for (int j = 0; j < TM; ++j)
                                if (j == jb + jj) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e] * p.alpha;
                                }
                        }
                        *reinterpret_cast<f32x4*>(Cs + (32 * jj + l31) * CS + wn0 + 32 * i + 8 * q + 4 * lhi) = v;
                    }
        }
        __syncthreads();
        if (p.ws)
            ep.store_partial(p, Cs, m0 + pass * WROWS, WROWS, split);
        else
            ep.store(p, Cs, m0 + pass * WROWS, WROWS);
        __syncthreads();
    }
}

template <int MODE, int BM, int BN, int NWM, int NWN, int BKT, int PJ, int WPE, int NS>
static int launch3(const GemmParams& p, uint32_t bA, uint32_t bA2, uint32_t bW, hipStream_t stream) {
    int tM = (p.M + BM - 1) / BM, tN = (p.N + BN - 1) / BN;
    size_t operands = (size_t)NS * (BM + BN) * BKT * 2;
    size_t staging = (size_t)(32 * PJ) * (BN + 4) * 4;
    size_t smem = operands > staging ? operands : staging;
    allow_big_smem(gemm3_kernel<MODE, BM, BN, NWM, NWN, BKT, PJ, WPE, NS>, smem);
    dim3 grid((unsigned)(((tM + 7) / 8) * 8 * tN), (unsigned)p.splits);
    MC_LAUNCH((gemm3_kernel<MODE, BM, BN, NWM, NWN, BKT, PJ, WPE, NS>), grid, dim3(64 * NWM * NWN), smem, stream, p, bA,
              bA2, bW, tM, tN);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

template <int MODE>
static int launch3_cfg(const GemmParams& p, uint32_t bA, uint32_t bA2, uint32_t bW, int cfg, hipStream_t s) {
    switch (cfg) {
        case 1: return launch3<MODE, 256, 320, 4, 2, 64, 2, 1, 2>(p, bA, bA2, bW, s);   // 8 waves, wave 64 x 160
        case 2: return launch3<MODE, 256, 256, 4, 2, 64, 2, 1, 2>(p, bA, bA2, bW, s);   // wave 64 x 128
        case 3: return launch3<MODE, 256, 128, 4, 2, 64, 2, 1, 2>(p, bA, bA2, bW, s);   // wave 64 x 64
        case 4: return launch3<MODE, 128, 320, 2, 2, 64, 2, 1, 2>(p, bA, bA2, bW, s);   // 4 waves, wave 64 x 160
        case 5: return launch3<MODE, 128, 256, 2, 4, 64, 2, 1, 2>(p, bA, bA2, bW, s);   // wave 64 x 64
        default: return MC_ERR_UNSUPPORTED;
    }
}

// cfg: see launch3_cfg.  Returns MC_ERR_UNSUPPORTED when an operand does not fit a 2 GiB buffer descriptor.
int gemm3_dispatch(const GemmParams& p, int mode, int cfg, size_t rowsA, hipStream_t stream) {
    size_t bytesA = (rowsA * (size_t)p.lda) * 2, bytesA2 = p.A2 ? (rowsA * (size_t)p.lda2) * 2 : 0;
    size_t bytesW = (size_t)p.N * p.K * 2;
    const size_t lim = 0x7FFFFFF0u;
    if (bytesA > lim || bytesA2 > lim || bytesW > lim) return MC_ERR_UNSUPPORTED;
    switch (mode) {
        case DENSE: return launch3_cfg<DENSE>(p, bytesA, bytesA2, bytesW, cfg, stream);
        case CONV_S1: return launch3_cfg<CONV_S1>(p, bytesA, bytesA2, bytesW, cfg, stream);
        case CONV_S2: return launch3_cfg<CONV_S2>(p, bytesA, bytesA2, bytesW, cfg, stream);
        case CONV_UP: return launch3_cfg<CONV_UP>(p, bytesA, bytesA2, bytesW, cfg, stream);
        default: return launch3_cfg<TCONV_S2>(p, bytesA, bytesA2, bytesW, cfg, stream);
    }
}

}  // namespace mc
