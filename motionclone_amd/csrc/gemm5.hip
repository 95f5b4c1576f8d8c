// GEMM / implicit-conv kernel, third generation of the 256x320 tile (contract: gemm_params.hpp, as gemm3.hip).
//
// What round 2 measured on gemm3: a k-step of the 2-stage loop takes ~4400 cycles against 2560 cycles of MFMA
// work per SIMD, because after every __syncthreads() both waves of a SIMD issue their 9 LDS-DMA instructions (~100 issue
// cycles each) at the same time and only then start their MFMAs, and the fp32 LDS staging epilogue (two waves writing,
// six waiting, four passes) costs ~7 us of a 40 us K = 640 tile.  This kernel keeps the tile (8 waves as 4 x 2, wave tile
// 64 x 160 of v_mfma_f32_32x32x16_f16, BN = 320 = the UNet's width quantum) and changes the two things that idle the pipe:
//
//  * K loop: a RING OF FOUR 32-deep operand stages (4 x 36 KiB), one raw s_barrier per stage, counted s_waitcnt vmcnt:
//    the loads of stage j+3 / j+4 are in flight while stage j is multiplied, nothing is drained at a barrier.  A stage is
//    two 16-deep k-slices; the barrier sits BETWEEN them: before it a wave has already fetched the fragments of slice 1,
//    after it it issues slice 1's MFMAs while the fragments of the next stage's slice 0 arrive - the matrix pipe has work
//    on both sides of every barrier.  The LDS-DMA issue is interleaved with the MFMAs one instruction at a time and
//    STAGGERED between the two waves of a SIMD: waves 0-3 (one per SIMD) issue theirs in the first half of a stage,
//    waves 4-7 in the second half, so a SIMD always has one wave that is only feeding the matrix pipe.
//  * Epilogue: every wave drains its own accumulators (no workgroup barrier, no idle waves): bias / alpha in fp32 in the
//    accumulator layout, fp16 through a WAVE-PRIVATE LDS image (row pitch 336 B: at most 2-way on the 8-byte transposing
//    writes), read back as whole 320-byte row segments and stored 16 bytes per lane with the residual added on the way
//    (R is rounded like the reference rounds it: Linear output to fp16, then the add - attention.py:293-299).
//
// Everything else is gemm3's: operand tiles global -> LDS by LDS-DMA with the XOR swizzle on the source address,
// hardware zero fill for conv padding / tails, K order of the convs (64-channel tile major, tap minor), XCD-aware tile
// order.  Split-K and odd shapes (N % 8, ldc % 8) stay on gemm3.
#include "gemm5_tile.hpp"

namespace mc {

// VAR (timing experiments, tools/gemm5_bench.py): bit 0 = STAGGER the LDS-DMA issue between the two waves of a SIMD (waves 0-3
// in the first half of a stage, waves 4-7 in the second; measured 1-8 % slower than everybody in the first half, which is the
// default), bit 1 = loads issued in one burst at the top of the half instead of interleaved with the MFMAs.
template <int MODE, int EPI, int VAR, int BM, int BN = g5::BN, int NW = g5::NW, int NS = g5::NS, int RES = 0, int GNS = 0>
__global__ __launch_bounds__(NW * 64, (NW == 8 || BN >= 256) ? 1 : 2) void gemm5_kernel(GemmParams p, uint32_t bytesA, uint32_t bytesA2,
                                                                          uint32_t bytesW, int tilesM, int tilesN, int sm, int sn) {
    using g5::BKT; using g5::RPI; using g5::STG; using g5::lds_off32;
    using T = g5::Tile<BM, BN, NW, NS>;
    constexpr int TN = T::TNV, WCOL = T::WCOL;
    constexpr int TM = T::TM, RA = T::RA, STAGE = T::STAGE, A_BYTES = T::A_BYTES, LA = T::LA, LB = T::LB, WB = T::WB, WX = T::WX, WXD = T::WXD;
    constexpr int WMW = T::WMW;
    MC_DYN_SMEM(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef MC_EMU
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: scalar branches on the wave group
#endif
    const bool grpA = wave < WX;     // the waves that move one more weight row group per stage
    const int pid = blockIdx.x;
    const int xcd = pid & 7, local = pid >> 3;   // workgroup b runs on XCD b % 8 (observed; a speed assumption only)
    int tm, tn;
    if (MODE == DENSE || tilesN > 1) {
        // Every XCD owns the M-tiles tm = 8 i + xcd and walks them in SUPER-TILES of sm x sn tiles (about the 32 workgroups
        // its CUs hold at a time): the activation rows of sm tiles and the weight rows of sn tiles are what the XCD's L2 serves
        // to those 32 tiles.  With one M-tile at a time (sm = 1: round 2's order) an XCD re-streams ALL of W for every
        // M-tile - PMC: 944 MB per launch against 131 MB algorithmic on the 8192 x 10240 x 1280 GEGLU layer (7.2x), 2.6-2.9x
        // on the other wide layers (profiles/r03_pmc_hbm_traffic.md).  sm, sn are chosen on the host (launch5).
        const int per_group = sm * tilesN;
        const int g = local / per_group, idx = local - g * per_group;
        const int blk = idx / (sm * sn), w = idx - blk * (sm * sn);
        tn = blk * sn + w / sm;
        tm = (g * sm + w % sm) * 8 + xcd;
    } else {
        // single-N-tile convolutions: a CONTIGUOUS range of M-tiles per XCD (halo rows of vertically adjacent tiles meet in L2)
        tn = 0;
        const int per = (tilesM + 7) >> 3;
        tm = local < per ? xcd * per + local : tilesM;
    }
    if (tm >= tilesM) return;
    const int m0 = tm * BM, n0 = tn * BN;

    const GBuf bufA = make_gbuf(p.A, bytesA);
    const GBuf bufA2 = make_gbuf(p.A2 ? p.A2 : p.A, p.A2 ? bytesA2 : bytesA);
    const GBuf bufW = make_gbuf(p.W, bytesW);

    // lane -> (row within the instruction's 16-row group, physical 16-byte slot); logical slot undoes the swizzle
    const int rsub = lane >> 2;
    const int lslot = (lane & 3) ^ (lane >> 4);

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
    // weight row groups: BN / 16 per stage; wave w takes w, w + NW, ... (WB of them) and (waves < WX) WB NW + w
    uint32_t w_off[WB + 1];
#pragma unroll
    for (int i = 0; i <= WB; ++i) {
        int n = n0 + (i < WB ? wave + NW * i : WB * NW + (wave % WXD)) * RPI + rsub;
        w_off[i] = n < p.N ? (uint32_t)n * (uint32_t)p.K * 2u + (uint32_t)lslot * 16u : kOOB;
    }

    // one LDS-DMA instruction of stage tile kt: pieces 0 .. RA-1 = activation row groups, then 2 (3) weight row groups
    auto issue_piece = [&](int kt, int buf, int piece) {
        char* base = smem + buf * STAGE;
        if (piece < RA) {
            const int i = piece;
            int tap = 0, c0 = kt * BKT;
            if (MODE != DENSE) {   // K order: 64-channel tile major, tap minor; a stage is half a 64-channel tile
                const int kt64 = kt >> 1;
                const int ct = kt64 / 9;
                tap = kt64 - 9 * ct;
                c0 = ct * 64 + (kt & 1) * 32;
            }
            const bool second = c0 >= p.c1;
            const int ld = second ? p.lda2 : p.lda;
            const int cc = (second ? c0 - p.c1 : c0) + lslot * 8;
            const int ky = tap / 3, kx = tap - 3 * (tap / 3);
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
            // out of range (padding, M tail): top bit set = beyond every descriptor = hardware zero fill; no branch
            uint32_t voff = (((uint32_t)row * (uint32_t)ld + (uint32_t)cc) * 2u) | (ok ? 0u : kOOB);
            char* dst = base + (wave + NW * i) * 1024;
            if (second)
                glds16(bufA2, voff, dst);
            else
                glds16(bufA, voff, dst);
        } else {
            const int i = piece - RA;
            uint32_t voff = w_off[i] + (uint32_t)kt * (BKT * 2u);   // kOOB + a small offset stays out of range
            glds16(bufW, voff, base + A_BYTES + (i < WB ? wave + NW * i : WB * NW + (wave % WXD)) * 1024);
        }
    };
    auto issue_stage = [&](int kt, int buf) {
#pragma unroll
        for (int q = 0; q < LB; ++q) issue_piece(kt, buf, q);
        if (grpA) issue_piece(kt, buf, LB);
    };
    // wait until at most `tiles` (0 .. NS - 2) of this wave's staged tiles are still in flight (loads retire in order)
    auto wait_tiles = [&](int tiles) {
        if (tiles <= 0) {
            wait_vmcnt_le<0>();
        } else if (tiles == 1 || NS == 3) {
            if (grpA) wait_vmcnt_le<LA>(); else wait_vmcnt_le<LB>();
        } else {
            if (grpA) wait_vmcnt_le<2 * LA>(); else wait_vmcnt_le<2 * LB>();
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wr = wave % WMW, wc = wave / WMW;
    const int wm0 = wr * (32 * TM), wn0 = wc * WCOL;
    const int l31 = lane & 31, lhi = lane >> 5;

    const int split = blockIdx.y;
    const int nk_all = p.K / BKT;
    const int kt_begin = (int)((long)split * nk_all / p.splits);
    const int nk = (int)((long)(split + 1) * nk_all / p.splits);

    // (Round 4, measured and removed: touching the tile's residual rows with one 4-byte LDS-DMA load per 128-byte line before the
    // first operand stage, so that the epilogue's lockstep read finds them in L2 / the Infinity Cache: every +R shape got
    // SLOWER inside the step loop - 32768 x 640 x 640 52.7 -> 57.0 us, 131072 x 320 x 1280 162 -> 175 us, the convs +3 % -
    // 34.6 vs 34.8 videos/min.  The rows compete with the operand stream for the same L2 while the k-loop runs.)
    // prologue: NS - 1 stages in flight (experiment VAR bit 0, ring of four only: waves 4-7 put a fourth one in flight - they
    // issue stage j + 4 in the second half of j)
    const bool stagger = NS == 4 && (VAR & 1) != 0;
    const int nst = nk - kt_begin;
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0)
        if (s0 < nst) issue_stage(kt_begin + s0, s0);
    if (stagger && !grpA && 3 < nst) issue_stage(kt_begin + 3, 3);

    half8_t fa[2][TM], fw[2][TN];   // slot 0: k-slice 0 of a stage, slot 1: k-slice 1
    auto load_frags = [&](int buf, int ks, half8_t* a, half8_t* w) {
        const char* bA = smem + buf * STAGE;
        const char* bW = bA + A_BYTES;
#pragma unroll
        for (int j = 0; j < TM; ++j)
            a[j] = *reinterpret_cast<const half8_t*>(bA + lds_off32(wm0 + 32 * j + l31, 2 * ks + lhi));
#pragma unroll
        for (int i = 0; i < TN; ++i)
            w[i] = *reinterpret_cast<const half8_t*>(bW + lds_off32(wn0 + 32 * i + l31, 2 * ks + lhi));
    };
    // Half a stage = one scheduling region: the 5 TM MFMAs of the current k-slice (`ca / cw`), the TM + 5 fragment reads of
    // the NEXT one (into `ra / rw`) and NL LDS-DMA instructions of stage `lkt` (ring slot `lbuf`), in this order:
    //   read a'0 [, a'1];  then for each weight fragment i:  MFMA (i,0) [, (i,1)], read w'i, [one LDS-DMA]
    // A weight fragment is dead after its MFMAs, so w'i can take its registers: 5 + 2 TM fragments live instead of
    // 2 x (5 + TM) (the 256-register budget holds 160 accumulators at TM = 2), and no instruction kind is issued in a burst.
    auto half_step = [&](auto nl_tag, auto p0_tag, int rbuf, int rks, half8_t* ra, half8_t* rw, const half8_t* ca, const half8_t* cw,
                         int lkt, int lbuf) {
        constexpr int NL = decltype(nl_tag)::value;      // LDS-DMA instructions woven into this half ...
        constexpr int P0 = decltype(p0_tag)::value;      // ... pieces P0 .. P0 + NL - 1 of the stage
        static_assert(NL <= TN, "one LDS-DMA behind each weight fragment at most");
        const char* bA = smem + rbuf * STAGE;
        const char* bW = bA + A_BYTES;
#pragma unroll
        for (int j = 0; j < TM; ++j)
            ra[j] = *reinterpret_cast<const half8_t*>(bA + lds_off32(wm0 + 32 * j + l31, 2 * rks + lhi));
#pragma unroll
        for (int i = 0; i < TN; ++i) {
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                if constexpr (TM == 4) {         // one wave per SIMD, 320 accumulators: column blocks 0-3 in AGPRs, 4 in VGPRs
                    if (i < 4) mfma32_agpr(acc[i][j], cw[i], ca[j]); else mfma32_vgpr(acc[i][j], cw[i], ca[j]);
                } else {
                    acc[i][j] = mfma32(cw[i], ca[j], acc[i][j]);
                }
            }
            rw[i] = *reinterpret_cast<const half8_t*>(bW + lds_off32(wn0 + 32 * i + l31, 2 * rks + lhi));
            if (i < NL) issue_piece(lkt, lbuf, P0 + i);
        }
#ifndef MC_EMU
        __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);          // DS read: a'0 [, a'1]
        if (NL > 0 && (VAR & 2)) __builtin_amdgcn_sched_group_barrier(0x010, NL, 0);   // experiment: LDS-DMA burst at the top
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);      // MFMA (i, 0) [, (i, 1)]
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // DS read w'i
            if (!(VAR & 2) && i < NL) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   // one LDS-DMA
        }
#endif
    };
    // a stage's LA / LB pieces: the first TN at most behind the weight fragments of the stage's first half, the rest (the
    // 4-wave geometry moves 6 / 7 per wave and stage) behind those of its second half
    constexpr int A1 = LA < TN ? LA : TN, A2 = LA - A1, B1 = LB < TN ? LB : TN, B2 = LB - B1;
    using N0 = std::integral_constant<int, 0>;
    using NA1 = std::integral_constant<int, A1>;
    using NA2 = std::integral_constant<int, A2>;
    using NB1 = std::integral_constant<int, B1>;
    using NB2 = std::integral_constant<int, B2>;
    using NLB = std::integral_constant<int, LB>;

    // stage 0 landed: this wave's own loads (counted: waves 4-7 may have three younger stages in flight), then everybody's
    {
        const int younger = min(stagger && !grpA ? 3 : NS - 2, nst - 1);
        if (younger >= 3) wait_vmcnt_le<3 * LB>(); else wait_tiles(younger);
    }
    raw_barrier();
    load_frags(0, 0, fa[0], fw[0]);

    // The loop is written out per wave group and per phase (steady state / tail) so that each body is one straight-line
    // scheduling region with a fixed instruction mix; every version executes exactly one barrier per stage.
    int buf = 0, kt = kt_begin;
    constexpr int PD = NS - 1;       // prefetch distance in stages
    auto nxt = [](int b) { return b + 1 == NS ? 0 : b + 1; };
    auto slot_of_new = [](int b) { return b == 0 ? NS - 1 : b - 1; };   // the slot freed at the previous barrier = (b + NS - 1) % NS
    if (grpA) {
        // steady state of the waves with LA pieces: stage kt + PD goes into the slot freed at the previous barrier
        for (; kt + PD < nk; ++kt) {
            const int nbuf = nxt(buf), lbuf = slot_of_new(buf);
            half_step(NA1(), N0(), buf, 1, fa[1], fw[1], fa[0], fw[0], kt + PD, lbuf);
            // stage kt + 1 landed (own loads: counted wait - NS - 3 younger stages and the pieces just issued stay in flight;
            // everybody's: barrier).  Past the barrier every wave has also finished reading stage kt (slice-1 fragments are
            // in registers)
            wait_vmcnt_le<(NS - 3) * LA + A1>();
            raw_barrier();
            half_step(NA2(), NA1(), nbuf, 0, fa[0], fw[0], fa[1], fw[1], kt + PD, lbuf);
            buf = nbuf;
        }
    } else if (!stagger) {
        // the other waves: the same with one weight row group less
        for (; kt + PD < nk; ++kt) {
            const int nbuf = nxt(buf), lbuf = slot_of_new(buf);
            half_step(NB1(), N0(), buf, 1, fa[1], fw[1], fa[0], fw[0], kt + PD, lbuf);
            wait_vmcnt_le<(NS - 3) * LB + B1>();
            raw_barrier();
            half_step(NB2(), NB1(), nbuf, 0, fa[0], fw[0], fa[1], fw[1], kt + PD, lbuf);
            buf = nbuf;
        }
    } else if constexpr (NS == 4 && LB <= TN) {
        // experiment (VAR bit 0, ring of four): waves 4-7 issue stage kt + 4 into slot `buf` right after the barrier that frees it
        for (; kt + 4 < nk; ++kt) {
            const int nbuf = nxt(buf);
            half_step(N0(), N0(), buf, 1, fa[1], fw[1], fa[0], fw[0], 0, 0);
            wait_vmcnt_le<2 * LB>();
            raw_barrier();
            half_step(NLB(), N0(), nbuf, 0, fa[0], fw[0], fa[1], fw[1], kt + 4, buf);
            buf = nbuf;
        }
    }
    // Round 6: the one-pass kernels leave through gemm6's epilogue (tile_epilogue: straight-line buffer loads / stores, bias through
    // an LDS strip, residual rows of pass p + 1 requested in front of the stores of pass p) whenever the tile's bias is ONE row -
    // no bias, a single row, or per-batch rows with rows_per_batch a multiple of the tile height.  The old epilogue waited
    // with vmcnt(0) for every predicated bias / residual load, i.e. for all stores issued before it: 9 us per 256x320 tile
    // with bias + residual (tools/tileloop_bench.py: attn_out_l1 46.0 us as used vs 33.3 plain; the new one 39.1 vs 35.4).
    // The wave's 160 bias values are requested here, in front of the last ring stages (the counted waits below only get stricter).
    const bool new_epi = !p.ws;      // (gemm5_dispatch admits only problems whose tiles see ONE bias row)
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (new_epi && p.bias) bias4 = load_bias4<WCOL>(p.bias, p.rows_per_batch >= p.M ? 0 : m0 / p.rows_per_batch, p.N, n0 + wn0);
    // tail: nothing left to issue (the next stage's slice-0 fragments read after the last stage are stale and never used)
    for (; kt < nk; ++kt) {
        const int nbuf = nxt(buf);
        half_step(N0(), N0(), buf, 1, fa[1], fw[1], fa[0], fw[0], 0, 0);
        wait_tiles(min(NS - 2, nk - 2 - kt));
        raw_barrier();
        half_step(N0(), N0(), nbuf, 0, fa[0], fw[0], fa[1], fw[1], 0, 0);
        buf = nbuf;
    }
    // every wave passed the last barrier after its final LDS read and no load is in flight: the ring is free

    if constexpr (TN != g5::TN) {
        if (p.ws) return;     // (gemm5_dispatch refuses split-K for the 128-column wave tiles: the slabs / reduce pass are 160-column)
    }
    if (p.ws) {
        // split-K: this workgroup's partial sums go to its slab in ACCUMULATOR-NATIVE order - chunk (i, j, q) of lane l at
        // float4 index ((i TM + j) 4 + q) 64 + l - so every store instruction writes 1 KiB contiguously; splitk_reduce5_kernel
        // sums the slabs in split order and runs the epilogue below
        float* slab = p.ws + ((((size_t)split * tilesM * tilesN + (size_t)tm * tilesN + tn) * NW + wave) * (size_t)(TN * TM * 16 * 64));
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
                    *reinterpret_cast<f32x4*>(slab + (((i * TM + j) * 4 + q) * 64 + lane) * 4) = v;
                }
        return;
    }
#ifdef MC_G5_OLD_EPILOGUE   // A/B build only (build.build_old_epilogue_control): round 5's epilogue
    if constexpr (TN == g5::TN) {
        g5_epilogue<EPI, TM>(p, acc, smem + wave * STG, m0 + wm0, n0 + wn0, lane);
        return;
    }
#endif
    if (new_epi) {
        EpiArgs e;
        e.C = p.C; e.R = p.R; e.M = p.M; e.N = p.N; e.ldc = p.ldc; e.ldr = p.ldr; e.alpha = p.alpha;
        const size_t outc = EPI == 1 ? (size_t)p.N / 2 : (size_t)p.N;
        e.bytesC = (uint32_t)(((size_t)(p.M - 1) * p.ldc + outc) * 2);
        e.bytesR = p.R ? (uint32_t)(((size_t)(p.M - 1) * p.ldr + outc) * 2) : 0u;
        e.has_bias = p.bias != nullptr;
        char* img = smem + wave * (32 * g5::RSG);                       // 32 rows x 80 columns per wave
        char* bstrip = smem + NW * (32 * g5::RSG) + wave * 640;         // 160 fp32 per wave
        if constexpr (GNS != 0) {
            e.gn_partial = p.gn_partial; e.gn_hw = p.gn_hw; e.gn_cpg = p.N / 32;
        }
        tile_epilogue<EPI, RES, TM, GNS, TN>(e, acc, img, bstrip, bias4, m0 + wm0, n0 + wn0);   // RES: with a residual (its own
        return;                                                          // instantiation: as a run-time branch the pair spilled)
    }
}

// Split-K second pass: one wave per wave tile.  Sums the `splits` slabs of its wave tile in split order (deterministic),
// then runs the same epilogue as the one-pass kernel.  grid = tiles * 8 single-wave workgroups: the reduction uses the
// whole chip's bandwidth however few tiles the problem has.
template <int TM>
__global__ __launch_bounds__(64) void splitk_reduce5_kernel(GemmParams p, int tilesM, int tilesN) {
    using namespace g5;
    MC_DYN_SMEM(smem);
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x / NW, wave = blockIdx.x % NW;
    const int tm = tile / tilesN, tn = tile - tm * tilesN;
    const int wr = wave & 3, wc = wave >> 2;
    constexpr int CH = TN * TM * 4;                              // float4 chunks per lane
    const size_t slab_floats = (size_t)CH * 64 * 4;
    const size_t split_stride = (size_t)tilesM * tilesN * NW * slab_floats;
    const float* base = p.ws + ((size_t)tile * NW + wave) * slab_floats + (size_t)lane * 4;
    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int s = 0; s < p.splits; ++s) {
        const float* src = base + (size_t)s * split_stride;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)(((i * TM + j) * 4 + q) * 64) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += v[e];
                }
    }
    g5_epilogue<0, TM>(p, acc, smem, tm * (128 * TM) + wr * (32 * TM), tn * BN + wc * 160, lane);
}

template <int MODE, int EPI, int VAR, int BM, int BN, int NW, int NS, int RES, int GNS = 0>
static int launch5r(const GemmParams& p, uint32_t bA, uint32_t bA2, uint32_t bW, hipStream_t stream) {
    using T = g5::Tile<BM, BN, NW, NS>;
    int tM = (p.M + BM - 1) / BM, tN = (p.N + BN - 1) / BN;
    allow_big_smem(gemm5_kernel<MODE, EPI, VAR, BM, BN, NW, NS, RES, GNS>, T::SMEM);
    // super-tile of the XCD-local tile order: sn divides the N-tiles, sm x sn ~ the workgroups an XCD holds at a time (32 CUs x
    // 1 or 2), least operand rows per tile
    const int rows_per_xcd = (tM + 7) / 8;
    const int resident = NW == 8 ? 32 : 64;
    int sm = 1, sn = 1;
    static const int no_swz = MC_ENV_INT("MC_GEMM5_NO_SUPERTILE", 0);   // A/B only
    // a weight matrix that fits the XCD's 4 MiB L2 beside the streaming activations is reused ACROSS M-tiles by the old order
    // (measured: qkv of level 1, 2.4 MB of weights, 100 -> 118 us with super-tiles); larger ones are re-streamed per M-tile
    const bool w_resident = (size_t)p.N * p.K * 2 <= (size_t)3 << 20;
    if ((MODE == DENSE || tN > 1) && !no_swz && !w_resident) {
        long best = -1;
        for (int c = 1; c <= tN && c <= 16; ++c) {
            if (tN % c) continue;
            int r = std::max(1, std::min(rows_per_xcd, (resident + c / 2) / c));
            long cost = ((long)r * BM + (long)c * BN) * 1000 / ((long)r * c);   // operand rows fetched per tile of the super-tile
            if (best < 0 || cost < best) best = cost, sm = r, sn = c;
        }
    }
    const int groups = (rows_per_xcd + sm - 1) / sm;
    dim3 grid((unsigned)(groups * sm * 8 * tN), (unsigned)p.splits);
    MC_LAUNCH((gemm5_kernel<MODE, EPI, VAR, BM, BN, NW, NS, RES, GNS>), grid, dim3(T::NTH), T::SMEM, stream, p, bA, bA2, bW, tM, tN, sm, sn);
    if (MC_LAST_ERROR()) return MC_ERR_LAUNCH;
    if constexpr (NW == 8 && BN == g5::BN) {   // (the split-K slabs / reduce pass are laid out for the 8-wave geometries)
        if (p.ws) {
            MC_LAUNCH((splitk_reduce5_kernel<T::TM>), dim3((unsigned)(tM * tN * g5::NW)), dim3(64), (size_t)g5::STG, stream, p, tM, tN);
            if (MC_LAST_ERROR()) return MC_ERR_LAUNCH;
        }
    }
    return MC_OK;
}

template <int MODE, int EPI, int VAR, int BM, int BN = g5::BN, int NW = g5::NW, int NS = g5::NS>
static int launch5(const GemmParams& p, uint32_t bA, uint32_t bA2, uint32_t bW, hipStream_t stream) {
    if constexpr (EPI == 0 && VAR == 0 && NW == 8 && BN == g5::BN && (MODE == DENSE || MODE == CONV_S1)) {
        if (p.gn_partial) {    // statistics of the output for the GroupNorm that reads it (gemm5_dispatch has checked the shape)
            if (p.R) return launch5r<MODE, EPI, VAR, BM, BN, NW, NS, 1, 1>(p, bA, bA2, bW, stream);
            return launch5r<MODE, EPI, VAR, BM, BN, NW, NS, 0, 1>(p, bA, bA2, bW, stream);
        }
    }
    if (p.gn_partial) return MC_ERR_UNSUPPORTED;
    if constexpr (EPI == 0) {
        if (p.R && !p.ws) return launch5r<MODE, EPI, VAR, BM, BN, NW, NS, 1>(p, bA, bA2, bW, stream);
    }
    return launch5r<MODE, EPI, VAR, BM, BN, NW, NS, 0>(p, bA, bA2, bW, stream);
}

// var: 0 = 256-row tiles, 8 waves; 4 = 128-row tiles; 5 = 256 x 160 tiles, 4 waves, ring of three: two workgroups per CU (dense
// only, no split-K); 6 = 256 x 320 tiles, FOUR waves with 128 x 160 wave tiles (one wave per SIMD, round 6; no split-K).
// (1 / 2 / 3 were round 4's schedule experiments - staggered / burst LDS-DMA issue, measured 1 - 8 % slower: the VAR template
// parameter still builds them, nothing instantiates them any more.)
template <int MODE>
static int launch5_var(const GemmParams& p, uint32_t bA, uint32_t bA2, uint32_t bW, int var, hipStream_t s) {
    if (var == 5) {
        if constexpr (MODE == DENSE) {
            if (p.ws || p.N % 160) return MC_ERR_UNSUPPORTED;
            if (p.epi == 1) return launch5<DENSE, 1, 0, 256, 160, 4, 3>(p, bA, bA2, bW, s);
            return launch5<DENSE, 0, 0, 256, 160, 4, 3>(p, bA, bA2, bW, s);
        }
        return MC_ERR_UNSUPPORTED;
    }
    if (var == 7) {      // 256 x 256 tiles, four waves, 128 x 128 wave tiles (dense, one pass)
        if constexpr (MODE == DENSE) {
            if (p.ws || p.gn_partial) return MC_ERR_UNSUPPORTED;
            return p.epi == 1 ? launch5<DENSE, 1, 0, 256, 256, 4, 4>(p, bA, bA2, bW, s) : launch5<DENSE, 0, 0, 256, 256, 4, 4>(p, bA, bA2, bW, s);
        }
        return MC_ERR_UNSUPPORTED;
    }
    if (var == 6) {
        if (p.ws) return MC_ERR_UNSUPPORTED;
        if (p.epi == 1) {
            if constexpr (MODE == DENSE) return launch5<DENSE, 1, 0, 256, 320, 4, 4>(p, bA, bA2, bW, s);
            return MC_ERR_UNSUPPORTED;
        }
        return launch5<MODE, 0, 0, 256, 320, 4, 4>(p, bA, bA2, bW, s);
    }
    if (p.epi == 1) {
        if (MODE != DENSE) return MC_ERR_UNSUPPORTED;
        return var == 4 ? launch5<DENSE, 1, 0, 128>(p, bA, bA2, bW, s) : launch5<DENSE, 1, 0, 256>(p, bA, bA2, bW, s);
    }
    if (var == 0) return launch5<MODE, 0, 0, 256>(p, bA, bA2, bW, s);
    if (var == 4) return launch5<MODE, 0, 0, 128>(p, bA, bA2, bW, s);
    return MC_ERR_UNSUPPORTED;
}

// Returns MC_ERR_UNSUPPORTED for what stays on gemm3: N / ldc / ldr not multiples of 8, operands >= 2 GiB.
// Split-K (p.ws != null, p.splits > 1): workspace of mc_workspace_bytes_gemm_splitk bytes, var 0 (256-row) or 4 (128-row tiles).
int gemm5_dispatch(const GemmParams& p, int mode, int var, size_t rowsA, hipStream_t stream) {
    size_t bytesA = (rowsA * (size_t)p.lda) * 2, bytesA2 = p.A2 ? (rowsA * (size_t)p.lda2) * 2 : 0;
    size_t bytesW = (size_t)p.N * p.K * 2;
    const size_t lim = 0x7FFFFFF0u;
    if (bytesA > lim || bytesA2 > lim || bytesW > lim) return MC_ERR_UNSUPPORTED;
    if ((p.ws != nullptr) != (p.splits > 1)) return MC_ERR_UNSUPPORTED;
    if (p.ws && (p.epi == 1 || (var != 0 && var != 4))) return MC_ERR_UNSUPPORTED;
    if (var == 5 && mode != DENSE) return MC_ERR_UNSUPPORTED;
    if (var == 1 || var == 2 || var == 3 || var > 7) return MC_ERR_UNSUPPORTED;
    if (var == 7 && mode != DENSE) return MC_ERR_UNSUPPORTED;
    if ((p.N & 7) || (p.ldc & 7) || (p.R && (p.ldr & 7)) || p.K % g5::BKT) return MC_ERR_UNSUPPORTED;
    if (p.epi == 1 && (p.N & 15)) return MC_ERR_UNSUPPORTED;
    if (p.gn_partial) {
        // GroupNorm statistics from the epilogue: one pass, whole groups per wave (160 columns), a wave tile inside one frame
        const int rows = var == 4 ? 32 : 64, cpg = p.N / 32;
        if (p.ws || p.epi || (var != 0 && var != 4) || (mode != DENSE && mode != CONV_S1)) return MC_ERR_UNSUPPORTED;
        if (p.N % 160 || p.N % 32 || (cpg != 10 && cpg != 20 && cpg != 40)) return MC_ERR_UNSUPPORTED;   // SD-1.5's 320 / 640 / 1280
        if (p.gn_hw <= 0 || p.gn_hw % rows || p.M % p.gn_hw) return MC_ERR_UNSUPPORTED;
    }
    if (!p.ws) {
        // the one-pass epilogue (tile_epilogue) reads ONE bias row per tile and addresses C / R through 2 GiB descriptors
        const int bm = var == 4 ? 128 : 256;
        if (p.bias && p.rows_per_batch < p.M && p.rows_per_batch % bm) return MC_ERR_UNSUPPORTED;
        if ((size_t)p.M * p.ldc * 2 > lim || (p.R && (size_t)p.M * p.ldr * 2 > lim)) return MC_ERR_UNSUPPORTED;
    }
    switch (mode) {
        case DENSE: return launch5_var<DENSE>(p, bytesA, bytesA2, bytesW, var, stream);
        case CONV_S1: return launch5_var<CONV_S1>(p, bytesA, bytesA2, bytesW, var, stream);
        case CONV_S2: return launch5_var<CONV_S2>(p, bytesA, bytesA2, bytesW, var, stream);
        case CONV_UP: return launch5_var<CONV_UP>(p, bytesA, bytesA2, bytesW, var, stream);
        default: return launch5_var<TCONV_S2>(p, bytesA, bytesA2, bytesW, var, stream);
    }
}

}  // namespace mc
