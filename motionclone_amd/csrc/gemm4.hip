// Streaming GEMM for the short-K Linear layers of the 64x64 level (K = 320: proj_in / proj_out, to_q|k|v, to_out,
// FeedForward's first Linear; reference attention.py:65,93,355-364, motion_module.py:113,135, diffusers FeedForward).
//
// Why a second structure: with K = 320 a 256x320 output tile of the tiled kernel (gemm3.hip) is only five k-steps; its
// prologue (cold operand fetch), the fp32 staged epilogue and the store tail take 4-5x the MFMA time of the tile, and
// with one 8-wave workgroup per CU nothing else runs meanwhile (measured: 350-550 TFLOP/s, 22 % MFMA utilisation).
// These layers are really streaming problems: A [M, 320] is read once, C [M, N] written once, W [N, 320] is tiny.
//
// Structure (one workgroup = 4 waves, 256 rows of A, two workgroups per CU):
//   * A-stationary in REGISTERS: a wave owns 64 rows; their MFMA B-operand fragments for the whole K (2 x K/16 x 16 B per
//     lane = 160 VGPRs at K = 320) are fetched once (global -> LDS by LDS-DMA in full lines, then ds_read_b128) and stay
//     in registers for the sweep over N.  No A traffic, no A staging instructions in the main loop.
//   * W streams through a 2-deep LDS ring in chunks of 32 output columns x K (20 KiB): 5 LDS-DMA instructions per wave per
//     chunk, one workgroup barrier per chunk, 40 MFMAs (v_mfma_f32_32x32x16_f16) + 20 ds_read_b128 per wave per chunk.
//   * the epilogue of a chunk is wave-private: accumulators -> fp16 -> the wave's own LDS staging rows -> read back as
//     16 rows x 64 B per instruction (+ the residual, which arrives through LDS-DMA as well) -> buffer stores.  No
//     barrier, no fp32 staging; the other workgroup on the CU computes meanwhile.
//   * bias: the accumulators START as the bias (a 1 KiB LDS-DMA per 8 chunks, ds_read_b128 broadcast), no epilogue add.
// Vector-memory counter discipline: vmcnt counts loads and stores, which retire out of order with respect to each other, so
// the per-chunk wait is vmcnt(0), placed after the chunk's MFMAs; the loads for later chunks (W two chunks ahead, residual
// rows, bias) are issued after the chunk's stores, which therefore had a whole chunk of MFMAs to drain when waited for.
#include "gemm_params.hpp"
#include "gn_stats.hpp"

namespace mc {

// Round 4: a normalisation of the rows of A applied IN REGISTERS during the A phase, so that the normalised tensor is never
// written to / read back from HBM (one launch and 4 T C bytes less per fused norm, at the level where the norms are HBM-bound):
//   kind 1  LayerNorm over the K = 320 channels (+ the temporal position table): the lane pair (l, l + 32) holds one row,
//           sum and sum of squares by v_dot2_f32_f16 (fp32) with one cross-lane add each, y = (x - mean) rstd gamma + beta
//           [+ pe[frame]] in fp32, rounded to fp16 once - what ln_fwd5_kernel (norm.hip) computes, i.e. attention.py:189,206,212 /
//           motion_module.py:204,210,237-246; (mean, rstd) optionally written for the backward.  The variance is
//           E[x^2] - mean^2 from fp32 sums (ln_fwd5_kernel and torch subtract the mean first): relative error ~1e-7 (1 +
//           mean^2 / var), i.e. it reaches the fp16 resolution of the output only for rows whose |mean| exceeds ~100 standard
//           deviations - GroupNorm's statistics (gn_finalize / gn_block_stats) have always been computed this way.  A second
//           pass over the register-resident fragments on (x - mean) (round-4 advice) would have to subtract in fp16 (the
//           fragments are the MFMA operands): each difference then carries up to 2^-11 of ITS size, a relative variance error
//           of ~1e-4 .. 5e-4 - worse than the one-pass fp32 formula for every row with |mean| < ~60 standard deviations; the
//           16-sigma test (tests/test_kernels.py) is where the two would still be ~30x apart in favour of this one;
//   kind 2  GroupNorm WITHOUT activation (Transformer3DModel.norm / TemporalTransformer3DModel.norm, eps 1e-6): per frame an
//           affine map x sc[k] + sh[k] (sc = rstd gamma, sh = beta - mean sc; the arithmetic of gn_apply_kernel); the
//           statistics are finalised from the per-chunk partial sums in the prologue (gn_block_stats) and written for the
//           backward by the first workgroup of every frame.  A workgroup's 256 rows lie in one frame (hw % 256 == 0).
constexpr int G4_BM = 256;          // rows per workgroup
constexpr int G4_BN = 32;           // output columns per chunk
constexpr int G4_WRING = 0;         // LDS map (bytes): W ring 2 x KS/4 x 4 KiB

template <int KS>
struct G4Lds {
    static constexpr int wchunk = (KS / 4) * 4096;       // [k-tile of 64][32 rows][128 B]
    static constexpr int land = 2 * wchunk;              // A landing ring: 2 x [256 rows][64 B]   (dead after the A phase)
    static constexpr int stg = land;                     // staging: 4 waves x [64 rows][128 B]    (same bytes as the landing ring)
    static constexpr int bias = stg + 4 * 8192;          // 2 x 1 KiB (256 floats = 8 chunks each)
    static constexpr int total = bias + 2048;
    static_assert(land + 2 * 16384 <= bias, "landing ring must end before the bias ring");
};

// Staging row of a wave: 128 B = the output columns of one STORE GROUP (plain: 2 chunks x 32 columns; GEGLU: 4 chunks x
// 16), as 8 slots of 16 B, slot s at physical slot s ^ ((row >> 1) & 7).  The image is lane-linear per 8 rows, so the
// residual rows of the group are DMA'd straight into it (the swizzle goes on the source address) and the accumulators are
// added IN PLACE; the group then leaves as whole 128-byte lines, 8 rows per store instruction.  Bank behaviour: the
// in-place 8-byte accesses are 2-way (the minimum for 32 rows x 8 B), the 16-byte read-back is conflict-free.
__device__ __forceinline__ int g4_stg_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

template <int KS, bool GEGLU, int NORM>
__global__ __launch_bounds__(256, 2) void gemm4_kernel(GemmParams p, uint32_t bytesA, uint32_t bytesW, uint32_t bytesC,
                                                        uint32_t bytesR, uint32_t bytesB, int tilesM, int nsplit,
                                                        int chunks, G4Norm np) {
    using L = G4Lds<KS>;
    constexpr int KT = KS / 4;          // 64-wide k-tiles of a W chunk
    constexpr int NT = KS / 2;          // 32-wide landing tiles of A
    MC_DYN_SMEM(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int pid = blockIdx.x;
    const int xcd = pid & 7, local = pid >> 3;
    const int ns = local % nsplit;
    const int tm = (local / nsplit) * 8 + xcd;
    if (tm >= tilesM) return;
    const int m0 = tm * G4_BM;
    const int c_begin = ns * chunks;                                    // first chunk of this workgroup's N range
    const int nchunks = min(chunks, p.N / G4_BN - c_begin);
    if (nchunks <= 0) return;

    const GBuf bufA = make_gbuf(p.A, bytesA);
    const GBuf bufW = make_gbuf(p.W, bytesW);
    const GBuf bufC = make_gbuf(p.C, bytesC);
    const GBuf bufR = make_gbuf(p.R ? (const void*)p.R : (const void*)p.C, p.R ? bytesR : 0u);
    const GBuf bufB = make_gbuf(p.bias ? (const void*)p.bias : (const void*)p.W, p.bias ? bytesB : 0u);
    char* const sW = smem + G4_WRING;
    char* const sLand = smem + L::land;
    char* const sStg = smem + L::stg + wave * 8192;
    char* const sBias = smem + L::bias;

    // ---- issue helpers ------------------------------------------------------------------------------------------------
    // W chunk c (absolute chunk index): this wave moves rows 8*wave .. +8 of every 64-wide k-tile
    const int wrow = 8 * wave + (lane >> 3);
    const int wslot = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);
    auto issue_w = [&](int c, int buf) {
        const int n = c * G4_BN + wrow;
        const uint32_t base = n < p.N ? ((uint32_t)n * (uint32_t)p.K + (uint32_t)wslot * 8u) * 2u : kOOB;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
            glds16(bufW, base == kOOB ? kOOB : base + (uint32_t)kt * 128u, sW + buf * L::wchunk + kt * 4096 + wave * 1024);
    };
    // residual rows of the store group starting at output column col0 (64 columns): 64 rows x 128 B into the staging image
    auto issue_r = [&](int col0, int lane) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = 8 * it + (lane >> 3);
            const int m = m0 + 64 * wave + row;
            const int lslot = (lane & 7) ^ ((row >> 1) & 7);
            const int col = col0 + lslot * 8;
            const uint32_t voff = (m < p.M && col < p.N) ? ((uint32_t)m * (uint32_t)p.ldr + (uint32_t)col) * 2u : kOOB;
            glds16(bufR, voff, sStg + it * 1024);
        }
    };
    // bias of 8 chunks (256 floats) starting at chunk group g of this workgroup's range
    auto issue_bias = [&](int g) {
        const uint32_t f = (uint32_t)((c_begin + 8 * g) * G4_BN + lane * 4);
        glds16(bufB, f < (uint32_t)p.N ? f * 4u : kOOB, sBias + (g & 1) * 1024);
    };

    // ---- A phase: the wave's 64 rows x K as MFMA B-operand fragments, resident for the whole sweep --------------------------
    half8_t a[2][KS];
    {
        if (wave == 0 && p.bias) issue_bias(0);
        issue_w(c_begin, 0);
        if (nchunks > 1) issue_w(c_begin + 1, 1);
        const int lslot = (lane & 3) ^ ((lane >> 4) & 3);
        auto issue_a = [&](int kt, int buf) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int g = wave + 4 * i;
                const int m = m0 + 16 * g + (lane >> 2);
                const uint32_t voff = m < p.M ? ((uint32_t)m * (uint32_t)p.lda + (uint32_t)(kt * 32 + lslot * 8)) * 2u : kOOB;
                glds16(bufA, voff, sLand + buf * 16384 + g * 1024);
            }
        };
        issue_a(0, 0);
        if (NT > 1) issue_a(1, 1);
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            if (kt + 1 < NT) wait_vmcnt_le<4>(); else wait_vmcnt_le<0>();
            raw_barrier();
            const char* b = sLand + (kt & 1) * 16384;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int row = 64 * wave + 32 * j + l31;
                    a[j][2 * kt + ks] = *reinterpret_cast<const half8_t*>(b + row * 64 + (((2 * ks + lhi) ^ ((row >> 2) & 3)) << 4));
                }
            raw_barrier();      // (waits lgkmcnt(0)): every wave has its fragments before the buffer is refilled
            if (kt + 2 < NT) issue_a(kt + 2, kt & 1);
        }
    }
    if (NORM != 0) {
        // The landing ring is dead (every wave holds its fragments: last barrier of the A phase) and the staging image that
        // shares its bytes is first written after the sweep's first barrier: room for the per-channel table
        //   tab[0 .. K) scale / gamma   tab[K .. 2K) shift / beta (+ position row)   tab[3K .. 3K + 64) group stats
        constexpr int KC = KS * 16;
        float* tab = reinterpret_cast<float*>(sLand);
        if (NORM == 2) {
            const int frame = m0 / np.hw;
            gn_block_stats(np.partial, frame, np.nchunk, np.gn_n, np.eps, 0, tab + 3 * KC, np.stats, ns == 0 && m0 % np.hw == 0);
            for (int k = tid; k < KC; k += 256) {
                const float* st = tab + 3 * KC + (k / (KC / 32)) * 2;
                const float sc = st[1] * np.gamma[k];
                tab[k] = sc;
                tab[KC + k] = np.beta[k] - st[0] * sc;
            }
        } else {
            const float* perow = np.pe ? np.pe + (size_t)((m0 / np.hw) % np.nframes_pe) * KC : nullptr;
            for (int k = tid; k < KC; k += 256) {
                tab[k] = np.gamma[k];
                tab[KC + k] = np.beta[k] + (perow ? perow[k] : 0.f);     // the position row enters with the shift
            }
        }
        __syncthreads();
        // kind 1: (mean, rstd) of the lane pair's row from ONE pass of v_dot2_f32_f16 (sum: x . (1, 1); sum of squares: x . x;
        // fp32 products and accumulation, var = E[x^2] - mean^2): 2 x 160 instructions per wave instead of the 1600 of a
        // convert-and-accumulate two-pass form, whose fp32 copies the compiler kept (and spilled)
        float nm[2] = {0.f, 0.f}, rs[2] = {1.f, 1.f};       // -mean * rstd, rstd
        if (NORM == 1) {
            half2_t ones;
            ones[0] = ones[1] = (half_t)1.0f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float su[4] = {0.f, 0.f, 0.f, 0.f}, sqv[4] = {0.f, 0.f, 0.f, 0.f};   // four independent chains each
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        half2_t v;
                        v[0] = a[j][s][2 * e];
                        v[1] = a[j][s][2 * e + 1];
                        su[e] = dot2acc(v, ones, su[e]);
                        sqv[e] = dot2acc(v, v, sqv[e]);
                    }
                float sum = (su[0] + su[1]) + (su[2] + su[3]), sq = (sqv[0] + sqv[1]) + (sqv[2] + sqv[3]);
                sum += shfl_xor(sum, 32);
                sq += shfl_xor(sq, 32);
                const float mean = sum / KC;
                const float rstd = 1.0f / sqrtf(fmaxf(sq / KC - mean * mean, 0.f) + np.eps);
                rs[j] = rstd;
                nm[j] = -mean * rstd;
                const int row = m0 + 64 * wave + 32 * j + l31;
                if (np.stats && ns == 0 && lhi == 0 && row < p.M) {
                    np.stats[(size_t)row * 2] = mean;
                    np.stats[(size_t)row * 2 + 1] = rstd;
                }
            }
        }
        // slice by slice, both row blocks of a slice from ONE read of the table (with the row blocks outermost hipcc keeps the
        // table of all 20 slices live across them - 320 floats - and spills 600-1300 registers).  Per element: one mixed-
        // precision FMA (xhat = x rstd - mean rstd, the fp16 operand read directly), one FMA (gamma, beta + pe), half a packed
        // convert; LayerNorm output cannot leave the fp16 range, so no saturation.  The affine map is one FMA + the convert.
        const float* tl = tab + 8 * opaque(lhi);
        f32x4 tg[2][2], tb[2][2];        // [parity of the slice][half]: the table of slice s + 1 is read while slice s is computed
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            tg[0][h] = *reinterpret_cast<const f32x4*>(tl + 4 * h);
            tb[0][h] = *reinterpret_cast<const f32x4*>(tl + KC + 4 * h);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + 1 < KS) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    tg[(s + 1) & 1][h] = *reinterpret_cast<const f32x4*>(tl + 16 * (s + 1) + 4 * h);
                    tb[(s + 1) & 1][h] = *reinterpret_cast<const f32x4*>(tl + 16 * (s + 1) + KC + 4 * h);
                }
            }
            float g[8], b[8];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e) g[4 * h + e] = tg[s & 1][h][e], b[4 * h + e] = tb[s & 1][h][e];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                half8_t o;
                if (NORM == 1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)(((float)a[j][s][e] * rs[j] + nm[j]) * g[e] + b[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = to_half((float)a[j][s][e] * g[e] + b[e]);
                }
#ifndef MC_EMU
                {   // the converted fragment has to EXIST here (register-only work is not ordered by a memory clobber)
                    u32x4 pin = __builtin_bit_cast(u32x4, o);
                    asm volatile("" : "+v"(pin));
                    o = __builtin_bit_cast(half8_t, pin);
                }
#endif
                a[j][s] = o;
            }
#ifndef MC_EMU
            __builtin_amdgcn_sched_barrier(0);     // one slice at a time (+ the next slice's table reads in flight)
#endif
        }
        // (the table is dead: the barrier below separates its last read from the first staging write)
    }
    if (!GEGLU && p.R) issue_r(c_begin * G4_BN, lane);
    wait_vmcnt_le<0>();
    raw_barrier();              // W chunks 0 / 1, bias group 0 and the first residual rows are in LDS for every wave

    // ---- sweep over the output columns ------------------------------------------------------------------------------
    // Per chunk: compute(c) from ring slot c&1 -> [own W(c+1), R(c) landed: vmcnt] -> ONE barrier (slot c&1 is free, every
    // wave's part of W(c+1) is visible) -> issue W(c+2) into the freed slot -> wave-private epilogue(c).  W is requested two
    // chunks ahead of its use and BEFORE the stores of the chunk in between, so no wait ever has a store in front of it.
    const float inv_alpha = 1.0f / p.alpha;
    const int wfrag_off = l31 * 128;            // + ((slot ^ ((l31 >> 1) & 7)) << 4) per slice
    const int wsw = (l31 >> 1) & 7;
#ifndef MC_EMU
    // PROFILING ONLY (dbg & 16): s_memtime stamps of the first 32 chunks, [workgroup][wave][chunk][6], through p.ws
    unsigned long long* stamps = (p.dbg & 16) && p.ws ? reinterpret_cast<unsigned long long*>(p.ws) + ((size_t)pid * 4 + wave) * 32 * 6 : nullptr;
#define G4_STAMP(k) if (stamps && ci < 32 && lane == 0) stamps[ci * 6 + (k)] = __builtin_readcyclecounter()
#else
#define G4_STAMP(k)
#endif
#pragma unroll 1
    for (int ci = 0; ci < nchunks; ++ci) {
        const int c = c_begin + ci;
        G4_STAMP(0);
        f32x16 acc[2];
        if (p.bias) {   // accumulators start as the bias: acc[r] belongs to column 8*(r>>2) + 4*lhi + (r&3)
            const float* bt = reinterpret_cast<const float*>(sBias + ((ci >> 3) & 1) * 1024) + (ci & 7) * G4_BN + 4 * lhi;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 bq = *reinterpret_cast<const f32x4*>(bt + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[0][4 * q + e] = acc[1][4 * q + e] = bq[e] * inv_alpha;   // out = alpha * acc
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
        }
        const char* bW = sW + (ci & 1) * L::wchunk + wfrag_off;
        half8_t wf[3];
        wf[0] = *reinterpret_cast<const half8_t*>(bW + (((0 + lhi) ^ wsw) << 4));
        wf[1] = *reinterpret_cast<const half8_t*>(bW + (((2 + lhi) ^ wsw) << 4));
        if (!(p.dbg & 2))
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            // fragment of slice s+2 requested before the MFMAs of slice s: three register sets, two reads in flight, so
            // the ds_read latency (~100+ cycles) runs under four MFMAs; the fences keep hipcc from sinking the read
            // back to its use (it otherwise folds the ring into one register set with lgkmcnt(0) per slice)
            if (s + 2 < KS)
                wf[(s + 2) % 3] = *reinterpret_cast<const half8_t*>(bW + ((s + 2) >> 2) * 4096 +
                                                                   (((2 * ((s + 2) & 3) + lhi) ^ wsw) << 4));
#ifndef MC_EMU
            __builtin_amdgcn_sched_barrier(0);
#endif
            acc[0] = mfma32(wf[s % 3], a[0][s], acc[0]);
            acc[1] = mfma32(wf[s % 3], a[1][s], acc[1]);
#ifndef MC_EMU
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        G4_STAMP(1);
        // Everything this wave has in flight must be down before the barrier: W(c+1), the residual rows, the bias - and the
        // previous chunk's stores.  vmcnt counts loads AND stores, and the two kinds retire out of order with respect to
        // each other, so a counted wait that leaves "the last n stores" pending can be satisfied while an older LDS-DMA is
        // still in flight (seen as rare wrong tiles once a second stream competed for the memory pipeline).  Hence
        // vmcnt(0), and the loads for later chunks are issued AFTER this chunk's stores: the wait then covers stores that
        // have had a whole chunk of MFMAs to drain.
        wait_vmcnt_le<0>();
        G4_STAMP(2);
        raw_barrier();
        G4_STAMP(3);
        const bool more_w = ci + 2 < nchunks && !(p.dbg & 8);
        const bool more_bias = wave == 0 && p.bias && (ci & 7) == 0 && 8 * ((ci >> 3) + 1) < nchunks;

        // ---- wave-private epilogue -----------------------------------------------------------------------------------
        if (p.dbg & 4) {
            if (more_w) issue_w(c + 2, ci & 1);
            if (more_bias) issue_bias((ci >> 3) + 1);
            continue;
        }
        // lane-derived addresses of the epilogue are re-derived here every chunk: hoisted out of the loop they would sit in
        // (or spill from) VGPRs that the resident A fragments need
        const int ln = opaque(lane);
        const int e31 = ln & 31, ehi = ln >> 5;
        constexpr int GRP = GEGLU ? 4 : 2;                 // chunks per 128-byte store group
        const int gi = ci % GRP;                           // position of this chunk inside its group (c_begin is a multiple of GRP)
        const bool flush = gi == GRP - 1 || ci + 1 == nchunks;
        if (GEGLU) {
            // W rows interleaved (h_j, gate_j): acc[4q+0], acc[4q+2] are h, acc[4q+1], acc[4q+3] their gates; the chunk
            // yields 16 output columns = staging slots 2*gi, 2*gi + 1
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    half2_t h;
                    h[0] = to_half(acc[j][4 * q + 0] * gelu_f(acc[j][4 * q + 1]));
                    h[1] = to_half(acc[j][4 * q + 2] * gelu_f(acc[j][4 * q + 3]));
                    const int col = 16 * gi + 4 * q + 2 * ehi;     // column inside the group, 2 halfs
                    *reinterpret_cast<half2_t*>(sStg + g4_stg_off(32 * j + e31, col >> 3) + (col & 7) * 2) = h;
                }
        } else {
            const bool scale = p.alpha != 1.0f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    half4_t h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = to_half(scale ? acc[j][4 * q + e] * p.alpha : acc[j][4 * q + e]);
                    char* dst = sStg + g4_stg_off(32 * j + e31, 4 * gi + q) + ehi * 8;
                    if (p.R) {   // the reference adds the residual to the layer's fp16 output in fp16 (attention.py:285-296)
                        const half4_t r = *reinterpret_cast<const half4_t*>(dst);
#pragma unroll
                        for (int e = 0; e < 4; ++e) h[e] = (half_t)((float)h[e] + (float)r[e]);
                    }
                    *reinterpret_cast<half4_t*>(dst) = h;
                }
        }
        if (!flush) {
            if (more_w) issue_w(c + 2, ci & 1);
            if (more_bias) issue_bias((ci >> 3) + 1);
            continue;
        }
        wave_lds_sync();
        // read-back + store in two halves of 32 rows (16 VGPRs of data in flight, the kernel sits at the 256-register
        // limit).  Vector-memory order of a flushing chunk: [stores x8][W(c+2) x5][bias][R x8]: every load the next
        // wait needs is YOUNGER than the stores, see above.
        const int grp = c / GRP;                           // absolute store-group index
        const int npieces = (gi + 1) * (8 / GRP);          // 16-byte pieces of the row that hold data (8 for a full group)
        half8_t o[4];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
                o[it] = *reinterpret_cast<const half8_t*>(sStg + g4_stg_off(32 * hh + 8 * it + (ln >> 3), ln & 7));
            wave_lds_sync();
            if (hh == 1) G4_STAMP(4);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int m = m0 + 64 * wave + 32 * hh + 8 * it + (ln >> 3);
                const uint32_t voff = (m < p.M && (ln & 7) < npieces) ? ((uint32_t)m * (uint32_t)p.ldc + (uint32_t)(grp * 64 + (ln & 7) * 8)) * 2u : kOOB;
                gbuf_st8(bufC, (p.dbg & 1) ? kOOB : voff, o[it]);
            }
        }
        if (more_w) issue_w(c + 2, ci & 1);
        if (more_bias) issue_bias((ci >> 3) + 1);
        // the staging image is in registers / on its way out: it may take the next group's residual rows
        if (!GEGLU && p.R && ci + 1 < nchunks) issue_r((grp + 1) * 64, ln);
        G4_STAMP(5);
    }
}

template <int KS, bool GEGLU, int NORM>
static int launch4(const GemmParams& p, uint32_t bA, uint32_t bW, uint32_t bC, uint32_t bR, uint32_t bB, int nsplit,
                   const G4Norm& np, hipStream_t stream) {
    const int tilesM = (p.M + G4_BM - 1) / G4_BM;
    const int nchunk_all = p.N / G4_BN;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > nchunk_all) nsplit = nchunk_all;
    int chunks = (nchunk_all + nsplit - 1) / nsplit;
    const int grp = GEGLU ? 4 : 2;                // chunks per 128-byte store group: ranges start at a group boundary
    chunks = (chunks + grp - 1) / grp * grp;
    nsplit = (nchunk_all + chunks - 1) / chunks;
    const size_t smem = G4Lds<KS>::total;
    allow_big_smem(gemm4_kernel<KS, GEGLU, NORM>, smem);
    dim3 grid((unsigned)(((tilesM + 7) / 8) * 8 * nsplit));
    MC_LAUNCH((gemm4_kernel<KS, GEGLU, NORM>), grid, dim3(256), smem, stream, p, bA, bW, bC, bR, bB, tilesM, nsplit, chunks, np);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

// DENSE, single A source, K = 320, N % 32 == 0, one bias row; optional residual, or the fused GEGLU epilogue (W rows
// interleaved (h_j, gate_j), C gets N/2 columns).  nsplit = workgroups sharing one 256-row block of A (0 = automatic).
// norm: null, or the normalisation applied to the rows of A in registers (G4Norm; no residual then).
// Returns MC_ERR_UNSUPPORTED for anything else (the caller falls back to the tiled kernels).
// Every reason gemm4_dispatch can refuse a problem, without launching anything: mc_norm_gemm_f16 asks BEFORE it launches the
// GroupNorm partial-sum pass, so that a refused call has done no work (its caller then runs norm and GEMM separately).
int gemm4_check(const GemmParams& p, const G4Norm* norm) {
    if (p.A2 || p.K != 320 || p.N % G4_BN || (p.ws && !(p.dbg & 16)) || p.splits != 1) return MC_ERR_UNSUPPORTED;
    if (p.epi && (p.R || p.alpha != 1.0f)) return MC_ERR_UNSUPPORTED;
    if (p.bias && p.rows_per_batch < p.M) return MC_ERR_UNSUPPORTED;
    if ((p.ldc & 7) || (p.R && (p.ldr & 7)) || (p.lda & 7)) return MC_ERR_UNSUPPORTED;
    const int kind = norm ? norm->kind : 0;
    if (kind) {
        if (p.R || kind < 1 || kind > 2 || !norm->gamma || !norm->beta) return MC_ERR_UNSUPPORTED;
        // the rows of a workgroup must lie in one frame wherever a per-frame quantity enters
        if ((kind == 2 || norm->pe) && (norm->hw <= 0 || norm->hw % G4_BM || p.M % norm->hw)) return MC_ERR_UNSUPPORTED;
        if (kind == 2 && (!norm->partial || !norm->stats || norm->nchunk <= 0)) return MC_ERR_UNSUPPORTED;
        if (kind == 1 && norm->pe && norm->nframes_pe <= 0) return MC_ERR_UNSUPPORTED;
    }
    const int ncol = p.epi ? p.N / 2 : p.N;
    const size_t lim = 0x7FFFFFF0u;
    const size_t bA = ((size_t)(p.M - 1) * p.lda + p.K) * 2, bW = (size_t)p.N * p.K * 2;
    const size_t bC = ((size_t)(p.M - 1) * p.ldc + ncol) * 2;
    const size_t bR = p.R ? ((size_t)(p.M - 1) * p.ldr + p.N) * 2 : 0;
    if (bA > lim || bW > lim || bC > lim || bR > lim) return MC_ERR_UNSUPPORTED;
    return MC_OK;
}

int gemm4_dispatch(const GemmParams& p, int nsplit, hipStream_t stream, const G4Norm* norm) {
    if (gemm4_check(p, norm) != MC_OK) return MC_ERR_UNSUPPORTED;
    const int kind = norm ? norm->kind : 0;
    const int ncol = p.epi ? p.N / 2 : p.N;
    const size_t bA = ((size_t)(p.M - 1) * p.lda + p.K) * 2, bW = (size_t)p.N * p.K * 2;
    const size_t bC = ((size_t)(p.M - 1) * p.ldc + ncol) * 2;
    const size_t bR = p.R ? ((size_t)(p.M - 1) * p.ldr + p.N) * 2 : 0, bB = (size_t)p.N * 4;
    if (nsplit <= 0) {   // two workgroups per CU want >= 512 of them; more splits re-read A from L2, fewer idle CUs
        const int tilesM = (p.M + G4_BM - 1) / G4_BM;
        nsplit = 1;
        while (tilesM * nsplit < 448 && (p.N / G4_BN) / (nsplit * 2) >= 5) nsplit *= 2;
    }
    const G4Norm none{};
    const G4Norm& np = norm ? *norm : none;
#define G4_GO(GE, NO) launch4<20, GE, NO>(p, (uint32_t)bA, (uint32_t)bW, (uint32_t)bC, (uint32_t)bR, (uint32_t)bB, nsplit, np, stream)
    if (p.epi) return kind == 0 ? G4_GO(true, 0) : (kind == 1 ? G4_GO(true, 1) : G4_GO(true, 2));
    return kind == 0 ? G4_GO(false, 0) : (kind == 1 ? G4_GO(false, 1) : G4_GO(false, 2));
#undef G4_GO
}

}  // namespace mc
