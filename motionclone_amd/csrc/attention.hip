// Flash-style spatial self-attention and text cross-attention, forward and
// data-gradient.  Replaces the reference's
// materialised [B*F*heads, N, N] score tensor (attention.py:461-490) and the
// xformers operator slot (attention.py:535-542): the score tile only ever exists in
// MFMA accumulators.
//
// Operands are read in place from token-major buffers ([tokens, 3C] fused q|k|v for
// self-attention; [tokens, C] queries + [B*77, 2C] text keys/values for
// cross-attention) through (pointer, row stride, head column offset) triples - the
// reference's reshape_heads_to_batch_dim copies (attention.py:367-379) are index math.
//
// Structure: 4 waves per workgroup, each owning 16*QT query rows (forward / dQ) or
// 16 key rows (dK/dV); the opposite operand streams through LDS in 64-row tiles
// (row-major copy + transposed copy where it is consumed as an MFMA "A" operand).
// As in temporal.hip the transposed score tile S^T = K Q^T is what is computed, so
// softmax statistics are per-lane scalars and P^T feeds the next MFMA as its B
// operand straight from the accumulator registers.
#include "mc_common.hpp"
#include <cstdlib>

namespace mc {

struct AParams {
    const half_t* q;
    const half_t* k;
    const half_t* v;
    int ldq, ldk, ldv;
    int Nq, Nk, heads, d, nbatch, kv_bdiv;
    float scale;
    int causal;   // forward only: key j attends to query i iff j <= i (CLIP text encoder); 0 on the UNet path
    int xcd_map;  // 1: 1-D grid, all row blocks of one (batch, head) on ONE XCD (attn_block); 0: grid (row blocks, heads, batch)
};

// Which (row block, head, batch element) this workgroup owns.  With the plain 3-D grid consecutive workgroups are the row
// blocks of ONE (batch, head) and the hardware deals them round-robin over the 8 XCDs, so every XCD's L2 fetches every K / V
// tile of every head (8x the fill traffic; the 16 row blocks of a level-0 head share 655 KB of K / V).  Here workgroup
// `pid` (on XCD pid % 8, a speed assumption only) takes unit u = (local / nxb) * 8 + xcd: all nxb row blocks of a unit run on
// the same XCD back to back and meet in its L2.
__device__ __forceinline__ bool attn_block(const AParams& P, int nxb, int& xb, int& h, int& b) {
    if (!P.xcd_map) {
        xb = blockIdx.x;
        h = blockIdx.y;
        b = blockIdx.z;
        return true;
    }
    const int pid = blockIdx.x, xcd = pid & 7, local = pid >> 3;
    const int u = (local / nxb) * 8 + xcd;
    xb = local - (local / nxb) * nxb;
    if (u >= P.heads * P.nbatch) return false;
    h = u % P.heads;
    b = u / P.heads;
    return true;
}
static inline dim3 attn_grid(const AParams& P, int nxb) {
    if (!P.xcd_map) return dim3((unsigned)nxb, (unsigned)P.heads, (unsigned)P.nbatch);
    const int units = P.heads * P.nbatch;
    return dim3((unsigned)(((units + 7) / 8) * 8 * nxb));
}

constexpr int KV_TILE = 64;
constexpr int TPAD = 72;  // row length (halfs) of transposed tiles: 64 + 8

// rows [r0, r0+64) x d of src (row stride ld) -> dst[64][RP] row-major, zero rows >= nrows
template <int DT>
__device__ __forceinline__ void stage_rows(half_t* dst, const half_t* src, size_t row_base, int r0, int nrows,
                                           int ld, int col0, int d) {
    constexpr int RP = DT * 16 + 8;
    const int vpr = d / 8;
    for (int idx = threadIdx.x; idx < KV_TILE * vpr; idx += blockDim.x) {
        int r = idx / vpr, vcol = idx - r * vpr;
        half8_t val = zero8();
        if (r0 + r < nrows) val = ld8(src + (row_base + r0 + r) * ld + col0 + vcol * 8);
        *reinterpret_cast<half8_t*>(dst + r * RP + vcol * 8) = val;
    }
}
// same rows, stored transposed: dst[c][r] (row length TPAD).  Each thread takes the same 8 columns of two adjacent
// rows and writes 8 packed b32 {row r, row r+1}: 32 lanes fill 32 consecutive dwords of one transposed row and the
// wave's two column groups sit 32 banks apart - a conflict-free transpose (scalar b16 writes were ~5-way conflicted)
__device__ __forceinline__ void stage_cols(half_t* dst, const half_t* src, size_t row_base, int r0, int nrows,
                                           int ld, int col0, int d) {
    const int vpr = d / 8;
    for (int pidx = threadIdx.x; pidx < (KV_TILE / 2) * vpr; pidx += blockDim.x) {
        const int vcol = pidx >> 5, r = 2 * (pidx & 31);
        const half_t* p = src + (row_base + r0 + r) * ld + col0 + vcol * 8;
        half8_t va = r0 + r < nrows ? ld8(p) : zero8();
        half8_t vb = r0 + r + 1 < nrows ? ld8(p + ld) : zero8();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            half2_t pr;
            pr[0] = va[e];
            pr[1] = vb[e];
            *reinterpret_cast<half2_t*>(dst + (vcol * 8 + e) * TPAD + r) = pr;
        }
    }
}
// zero the padding columns [d, DT*16) of a row-major tile and padding rows of a transposed tile
template <int DT>
__device__ __forceinline__ void zero_pads(half_t* rows_tile, half_t* cols_tile, int d) {
    constexpr int RP = DT * 16 + 8;
    const int padc = DT * 16 - d;
    if (rows_tile)
        for (int idx = threadIdx.x; idx < KV_TILE * padc; idx += blockDim.x) {
            int r = idx / padc, c = d + idx % padc;
            rows_tile[r * RP + c] = (half_t)0.f;
        }
    if (cols_tile)
        for (int idx = threadIdx.x; idx < padc * TPAD; idx += blockDim.x) cols_tile[d * TPAD + idx] = (half_t)0.f;
}

__device__ __forceinline__ float grp_max(float v) {
    v = fmaxf(v, shfl_xor(v, 16));
    return fmaxf(v, shfl_xor(v, 32));
}
__device__ __forceinline__ float grp_sum(float v) {
    v += shfl_xor(v, 16);
    return v + shfl_xor(v, 32);
}

// ---- forward ----------------------------------------------------------------------------------
// K/V tiles are double-buffered in LDS; the next tile's global loads are issued into registers before the current
// tile's MFMAs and written to LDS after them (one barrier per tile).  Pairs of K=16 steps run as one
// v_mfma_f32_16x16x32_f16 (cat4: concatenated fragments), the odd step of d = 40 / 80 as the K=16 instruction.
// (second launch-bound = minimum waves per SIMD: without it hipcc budgets 512 registers, parks the MFMA accumulators in
// AGPRs and pays v_accvgpr_read/write VALU slots around every softmax - 168 of them per tile at d = 40)
template <int DT, int QT, bool PF>
__global__ __launch_bounds__(256, (QT == 1 && DT <= 4) ? 4 : 2) void attn_fwd_kernel(AParams P, half_t* o, int ldo, float* lse) {
    constexpr int RP = DT * 16 + 8;
    constexpr int KS_HALFS = KV_TILE * RP;
    constexpr int VT_HALFS = DT * 16 * TPAD;
    constexpr int NV = (KV_TILE * DT * 2 + 255) / 256;  // 16-byte vectors per thread per staged matrix
    MC_DYN_SMEM(smem);
    constexpr int NB = PF ? 2 : 1;                      // LDS buffers per matrix
    half_t* Ks0 = reinterpret_cast<half_t*>(smem);      // [NB][64][RP]
    half_t* Vt0 = Ks0 + NB * KS_HALFS;                  // [NB][DT*16][TPAD]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c15 = lane & 15;
    int xb, h, b;
    if (!attn_block(P, (P.Nq + 64 * QT - 1) / (64 * QT), xb, h, b)) return;
    const int col0 = h * P.d;
    const size_t qbase = (size_t)b * P.Nq, kbase = (size_t)(b / P.kv_bdiv) * P.Nk;
    const int q0 = xb * (64 * QT) + wave * (16 * QT);
    const int vpr = P.d / 8;
    const float sl2 = P.scale * 1.4426950408889634f;  // scores are carried in log2 units

    zero_pads<DT>(Ks0, Vt0, P.d);
    if (PF) zero_pads<DT>(Ks0 + KS_HALFS, Vt0 + VT_HALFS, P.d);
    // head dims that leave padding rows in V^T (d = 40 -> 48): row d is set to ones, so O^T row d accumulates the
    // softmax denominators inside the PV MFMAs (rescaled with the rest of the accumulator), off the VALU
    const bool ones_row = P.d < DT * 16;
    if (ones_row) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < TPAD; idx += blockDim.x) {
            Vt0[P.d * TPAD + idx] = (half_t)1.0f;
            if (PF) Vt0[VT_HALFS + P.d * TPAD + idx] = (half_t)1.0f;
        }
    }

    // per-thread staging slots; only the base row advances.  K: fixed (row, 16-byte column) of the 64-row tile, written
    // row-major as one b128.  V: fixed (row PAIR, 16-byte column): the thread holds the same 8 columns of two adjacent
    // kv rows and writes the transposed image as 8 packed b32 {V[r][c], V[r+1][c]} - 32 lanes cover 32 consecutive
    // dwords of one V^T row and the two column groups of a wave land 32 banks apart, so the transpose costs no LDS
    // bank conflicts (the former 16 ds_write_b16 per thread kept the LDS pipe ~75 % busy, over half of it conflicts)
    constexpr int NVP = (KV_TILE / 2 * DT * 2 + 255) / 256;
    half8_t rk[NV], rva[NVP], rvb[NVP];
    bool s_ok[NV], v_ok[NVP];
    int s_lk[NV], v_l[NVP];
    uint32_t s_ok_off[NV], v_off[NVP];   // byte offsets into this batch's K / V rows (hardware range check: rows >= Nk -> 0)
    const GBuf kbuf = make_gbuf(P.k + kbase * P.ldk, (uint32_t)P.Nk * P.ldk * 2);
    const GBuf vbuf = make_gbuf(P.v + kbase * P.ldv, (uint32_t)P.Nk * P.ldv * 2);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int idx = threadIdx.x + 256 * i;
        int r = idx / vpr, vcol = idx - r * vpr;
        s_ok[i] = idx < KV_TILE * vpr;
        s_lk[i] = r * RP + vcol * 8;
        s_ok_off[i] = (uint32_t)(r * P.ldk + col0 + vcol * 8) * 2;
    }
#pragma unroll
    for (int i = 0; i < NVP; ++i) {
        int pidx = threadIdx.x + 256 * i;
        int vcol = pidx >> 5, rp = pidx & 31;
        v_ok[i] = vcol < vpr;
        v_l[i] = vcol * 8 * TPAD + 2 * rp;
        v_off[i] = (uint32_t)(2 * rp * P.ldv + col0 + vcol * 8) * 2;
    }
    auto load_regs = [&](int kv0) {
        const uint32_t ko = (uint32_t)kv0 * P.ldk * 2, vo = (uint32_t)kv0 * P.ldv * 2;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (s_ok[i]) rk[i] = gbuf_ld8(kbuf, s_ok_off[i] + ko);
#pragma unroll
        for (int i = 0; i < NVP; ++i)
            if (v_ok[i]) {
                rva[i] = gbuf_ld8(vbuf, v_off[i] + vo);
                rvb[i] = gbuf_ld8(vbuf, v_off[i] + vo + (uint32_t)P.ldv * 2);
            }
    };
    auto store_regs = [&](int buf) {
        half_t* Ks = Ks0 + buf * KS_HALFS;
        half_t* Vt = Vt0 + buf * VT_HALFS;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (s_ok[i]) *reinterpret_cast<half8_t*>(Ks + s_lk[i]) = rk[i];
#pragma unroll
        for (int i = 0; i < NVP; ++i)
            if (v_ok[i]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    half2_t pr;
                    pr[0] = rva[i][e];
                    pr[1] = rvb[i][e];
                    *reinterpret_cast<half2_t*>(Vt + v_l[i] + e * TPAD) = pr;
                }
            }
    };

    half4_t qf[QT][DT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        int qi = q0 + 16 * t + c15;
#pragma unroll
        for (int ks = 0; ks < DT; ++ks) {
            int c = 16 * ks + 4 * g;
            qf[t][ks] = (qi < P.Nq && c < P.d) ? ld4(P.q + (qbase + qi) * P.ldq + col0 + c) : zero4();
        }
    }
    f32x4 oacc[QT][DT];
    float m[QT], l[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m[t] = -INFINITY;
        l[t] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) oacc[t][dt] = fzero4();
    }

    if (PF) {
        load_regs(0);
        __syncthreads();  // pad zeroing done
        store_regs(0);
        __syncthreads();
    }
    int buf = 0;
    for (int kv0 = 0; kv0 < P.Nk; kv0 += KV_TILE) {
        const bool more = kv0 + KV_TILE < P.Nk;
        if (PF) {
            if (more) load_regs(kv0 + KV_TILE);   // in flight during this tile's MFMAs
        } else {
            __syncthreads();                      // previous tile's reads done (and pad zeroing, first time)
            load_regs(kv0);
            store_regs(0);
            __syncthreads();
        }
        const half_t* Ks = Ks0 + buf * KS_HALFS;
        const half_t* Vt = Vt0 + buf * VT_HALFS;

        f32x4 st[QT][4];   // first MFMA of each tile takes the inline-constant zero as C: no accumulator clearing
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const half_t* krow = Ks + (16 * j + c15) * RP + 4 * g;
#pragma unroll
            for (int ks = 0; ks + 1 < DT; ks += 2) {
                half8_t kf = cat4(ld4(krow + 16 * ks), ld4(krow + 16 * ks + 16));
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    st[t][j] = mfma16k32(kf, cat4(qf[t][ks], qf[t][ks + 1]), ks == 0 ? fzero4() : st[t][j]);
            }
            if (DT & 1) {
                // odd step: same K=32 instruction with the upper k-slots zero.  (A dependent legacy 16x16x16 MFMA
                // issued right behind 16x16x32 ones on the same accumulator returned wrong sums on gfx950.)
                half8_t kf = cat4(ld4(krow + 16 * (DT - 1)), zero4());
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    st[t][j] = mfma16k32(kf, cat4(qf[t][DT - 1], zero4()), DT == 1 ? fzero4() : st[t][j]);
            }
        }
        // online softmax in the exp2 domain.  VALU is the bound of this loop (a wave64 VALU op holds the SIMD for 4
        // cycles, v_exp_f32 for 16), so per score: one max on the RAW score, one FMA (score * sl2 - m), one exp2, half a
        // packed convert.  Row sums come out of the PV MFMAs when the head dim leaves a padding row in V^T (ones row).
        half8_t pf[QT][2];
        const bool tail = kv0 + KV_TILE > P.Nk;
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            if (tail) {  // wave-uniform branch: only the ragged last tile pays for masking
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (kv0 + 16 * j + 4 * g + i >= P.Nk) st[t][j][i] = -INFINITY;
            }
            if (P.causal) {  // wave-uniform: the lane's query row is q0 + 16 t + c15, its keys kv0 + 16 j + 4 g + i
                const int qrow = q0 + 16 * t + c15;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (kv0 + 16 * j + 4 * g + i > qrow) st[t][j][i] = -INFINITY;
            }
            // ablation builds (tools/attn_ablate.py; never defined in the product build): -DMC_ATTN_NOMAX drops the running-max
            // reduction, -DMC_ATTN_NOEXP the exponential, -DMC_ATTN_NOSOFTMAX everything between the two MFMA groups but the
            // fp16 conversion - wrong results, they only price the softmax's VALU work against the MFMA work
            float mx = max3(st[t][0][0], st[t][0][1], st[t][0][2]);
            mx = max3(mx, st[t][0][3], st[t][1][0]);
            mx = max3(mx, st[t][1][1], st[t][1][2]);
            mx = max3(mx, st[t][1][3], st[t][2][0]);
            mx = max3(mx, st[t][2][1], st[t][2][2]);
            mx = max3(mx, st[t][2][3], st[t][3][0]);
            mx = max3(mx, st[t][3][1], st[t][3][2]);
            mx = fmaxf(mx, st[t][3][3]);
            mx = grp_max(mx);
#if defined(MC_ATTN_NOMAX) || defined(MC_ATTN_NOSOFTMAX)
            mx = 0.f;
#endif
            const float mnew = fmaxf(m[t], mx * sl2);   // sl2 > 0: max commutes with the scaling
            float rs = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
#if defined(MC_ATTN_NOSOFTMAX)
                    float e0 = st[t][j][i], e1 = st[t][j][i + 1];
#elif defined(MC_ATTN_NOEXP)
                    float e0 = fmaf(st[t][j][i], sl2, -mnew), e1 = fmaf(st[t][j][i + 1], sl2, -mnew);
#else
                    float e0 = fast_exp2(fmaf(st[t][j][i], sl2, -mnew));
                    float e1 = fast_exp2(fmaf(st[t][j][i + 1], sl2, -mnew));
#endif
                    if (!ones_row) rs += e0 + e1;
                    half2_t h2 = pk_rtz(e0, e1);
                    pf[t][j >> 1][4 * (j & 1) + i] = h2[0];
                    pf[t][j >> 1][4 * (j & 1) + i + 1] = h2[1];
                }
            if (!ones_row) rs = grp_sum(rs);
            if (wave_all(mnew == m[t])) {
                l[t] += rs;
            } else {
                const float alpha = fast_exp2(m[t] - mnew);
                l[t] = l[t] * alpha + rs;
                m[t] = mnew;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) oacc[t][dt][i] *= alpha;
            }
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const half_t* vrow = Vt + (16 * dt + c15) * TPAD + 4 * g;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                half8_t vf = cat4(ld4(vrow + 32 * jp), ld4(vrow + 32 * jp + 16));
#pragma unroll
                for (int t = 0; t < QT; ++t) oacc[t][dt] = mfma16k32(vf, pf[t][jp], oacc[t][dt]);
            }
        }
        if (PF) {
            if (more) store_regs(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }

#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = q0 + 16 * t + c15;
        // O^T row d sits in accumulator tile DT-1, element 0 of lane group (d % 16) / 4 (d is a multiple of 8)
        const float lt = ones_row ? shfl(oacc[t][DT - 1][0], 16 * ((P.d & 15) >> 2) + c15) : l[t];
        if (qi >= P.Nq) continue;
        const float inv = 1.0f / lt;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            int c = 16 * dt + 4 * g;
            if (c < P.d) {
                half4_t ov;
#pragma unroll
                for (int i = 0; i < 4; ++i) ov[i] = to_half(oacc[t][dt][i] * inv);
                st4(o + (qbase + qi) * ldo + col0 + c, ov);
            }
        }
        if (g == 0 && lse) lse[((size_t)b * P.heads + h) * P.Nq + qi] = (m[t] + log2f(lt)) * 0.6931471805599453f;
    }
}

// ---- forward, long sequences at d = 40: K / V through an LDS-DMA ring, softmax offset inside the MFMA ---------------------
// Same mathematics as attn_fwd_kernel (S^T = K Q^T tile by tile, P^T as the B operand of the PV MFMAs straight from the
// accumulators, fp32 statistics); everything around the MFMAs is different, because the old loop was bound by what sat BETWEEN
// them (VALU 58 % / MFMA 26 % of the resident cycles, 44 % of the wave time waiting: profiles/r03_pmc_sq_gemm_attention.md):
//   * K and V tiles (64 keys) arrive by LDS-DMA (buffer_load ... lds) into a ring of three stages, two tiles ahead of their
//     use: no staging registers, no VALU, no exposed load latency, ONE bare barrier per tile (the old loop had two, with the
//     global load between them);
//   * both images are row-major [64][32 DT bytes].  K fragments are 16-byte reads (the contraction index is permuted so a lane's
//     8 k-slots are 8 adjacent head-dim elements; chunks of rows 8..15 of every 16 are swapped pairwise, which makes the reads
//     conflict-free); V^T fragments come from the SAME row-major layout through the hardware transpose read
//     (ds_read_b64_tr_b16; the 24-dword pitch is conflict-free for it) - the register transpose and its packed writes are gone;
//   * head-dim padding is hardware zero fill: the chunk [d, 16 DT) of every row is fetched out of range.  The ones row that
//     makes the PV MFMAs produce the softmax denominators is OR-ed into the one lane's V^T fragment that holds row d;
//   * THE SOFTMAX OFFSET RIDES IN A PADDING K-SLOT.  Q is pre-multiplied by scale * log2(e) when its fragments are loaded, one
//     padding slot of the head dim holds 1.0 on the K side and -m[q] on the Q side, so the MFMA returns s * scale * log2(e) -
//     m[q] and a score costs ONE v_exp_f32 and half a packed convert - no FMA, no running-max update.  m[q] is an offset, not
//     the maximum: it only has to keep 2^(s - m) inside fp16, so it is re-based (output and denominators rescaled, as in any
//     online softmax) only when a score of the tile exceeds it by more than 2^8 - the first tile always, a handful of times
//     after.  The detection is a per-lane compare of the lane's own 16 scores (no cross-lane reduction on the common path);
//     the re-base takes the row maximum with two VALU lane swaps (rows_max), not ds_bpermute round trips.  Because the SAME
//     fp16 offset multiplies every key of a row, its rounding cancels in the normalisation;
//   * per query tile the order is softmax(t) -> PV(t), so the VALU of tile t + 1 issues under the MFMAs of tile t.
// The 16-wide remainder of the head dim runs as a zero-extended 32-wide step: v_mfma_f32_16x16x16_f16 occupies the pipe for the
// same 16 cycles (tools/issue_rate.py), and mixed with 16x16x32 on one accumulator it returned wrong sums on gfx950 (again in
// this kernel: 5 of 18 GPU tests failed with it, the simulator passed).
#ifndef MC_ATTN_DKDV_KT
#define MC_ATTN_DKDV_KT 4   // key tiles (16 rows) per wave in attn_bwd_dkdv_ring_kernel at d = 40
#endif
constexpr float kRebase = 8.0f;   // re-base a row's offset when 2^(score - offset) would pass 2^8
// PADROW: the head dim leaves a padding row in V^T (d = 40) that becomes the ones row; otherwise (d = 80) the denominators take
// one more output tile whose V^T fragment is a constant (row 0 = ones): two more MFMAs per query tile instead of 16 adds.
template <int DT, int QT, bool PADROW>
__global__ __launch_bounds__(256, (DT == 3 && QT == 2) ? 4 : 2) void attn_fwd_ring_kernel(AParams P, half_t* o, int ldo, float* lse) {
    constexpr int DTO = DT + (PADROW ? 0 : 1);   // output tiles incl. the denominators' row
    constexpr int CPR = 2 * DT;          // 16-byte chunks per image row
    constexpr int PB = 32 * DT;          // row pitch, bytes
    constexpr int IMG = KV_TILE * PB;    // one matrix of one stage
    constexpr int STAGE = 2 * IMG;
    constexpr int NSTG = 3;
    constexpr int IPW = 2 * CPR / 4;     // LDS-DMA instructions per wave and tile (K + V = 2 CPR KiB over 4 waves)
    constexpr int NMAIN = DT / 2;        // 32-wide steps of the head dim
    static_assert(DT & 1, "needs the 16-wide remainder step: its padding k-slots carry the softmax offset");
    static_assert((2 * CPR) % 4 == 0, "stage must split evenly over the four waves");
    MC_DYN_SMEM(smem);
    const int lane = threadIdx.x & 63;
#ifdef MC_EMU
    const int wave = threadIdx.x >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably uniform: K / V descriptor choice is scalar
#endif
    const int g = lane >> 4, c15 = lane & 15;
    int xb, h, b;
    if (!attn_block(P, (P.Nq + 64 * QT - 1) / (64 * QT), xb, h, b)) return;
    const int col0 = h * P.d;
    const size_t qbase = (size_t)b * P.Nq, kbase = (size_t)(b / P.kv_bdiv) * P.Nk;
    const int q0 = xb * (64 * QT) + wave * (16 * QT);
    const int vpr = P.d / 8;
    const float sl2 = P.scale * 1.4426950408889634f;
    const int nk = (P.Nk + KV_TILE - 1) / KV_TILE;

    // ---- LDS-DMA slots of this lane: instruction n = wave + 4 i fills chunks [64 (n % CPR), +64) of K (n < CPR) or V
    const GBuf kbuf = make_gbuf(P.k + kbase * P.ldk, (uint32_t)P.Nk * P.ldk * 2);
    const GBuf vbuf = make_gbuf(P.v + kbase * P.ldv, (uint32_t)P.Nk * P.ldv * 2);
    uint32_t dma_off[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int n = wave + 4 * i;
        const bool is_v = n >= CPR;
        const int pch = (n % CPR) * 64 + lane;        // chunk position inside the image
        const int row = pch / CPR, cp = pch % CPR;
        const int c = is_v ? cp : (cp ^ ((row >> 3) & 1));
        const int ld = is_v ? P.ldv : P.ldk;
        dma_off[i] = c < vpr ? (uint32_t)(row * ld + col0 + 8 * c) * 2 : kOOB;
    }
    auto issue = [&](int tile) {
        char* stg = smem + (tile % NSTG) * STAGE;
        const uint32_t ko = (uint32_t)tile * KV_TILE * P.ldk * 2, vo = (uint32_t)tile * KV_TILE * P.ldv * 2;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int n = wave + 4 * i;
            if (n >= CPR) glds16(vbuf, dma_off[i] + vo, stg + IMG + (n - CPR) * 1024);
            else glds16(kbuf, dma_off[i] + ko, stg + n * 1024);
        }
    };

    // ---- Q fragments (B operand), pre-multiplied by scale * log2(e): 32-wide step s holds head-dim [32 s + 8 g, +8), the
    // remainder step [32 NMAIN + 4 g, +4) followed by 4 padding slots; the offset slot is k-slot 4 of lane group 0.
    half8_t qm[QT][NMAIN > 0 ? NMAIN : 1];
    half4_t qr[QT], qx[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = q0 + 16 * t + c15;
        const half_t* qp = P.q + (qbase + qi) * P.ldq + col0;
#pragma unroll
        for (int s2 = 0; s2 < NMAIN; ++s2) {
            qm[t][s2] = qi < P.Nq ? ld8(qp + 32 * s2 + 8 * g) : zero8();
#pragma unroll
            for (int e = 0; e < 8; ++e) qm[t][s2][e] = (half_t)((float)qm[t][s2][e] * sl2);
        }
        qr[t] = (qi < P.Nq && 32 * NMAIN + 4 * g < P.d) ? ld4(qp + 32 * NMAIN + 4 * g) : zero4();
#pragma unroll
        for (int e = 0; e < 4; ++e) qr[t][e] = (half_t)((float)qr[t][e] * sl2);
        qx[t] = zero4();
    }
    issue(0);
    if (nk > 1) issue(1);
#ifndef MC_EMU
    // a use of the Q registers HERE: hipcc then waits for the Q loads (older than the two tiles in flight, so a counted wait)
    // in front of the loop instead of putting a vmcnt(0) - which would also drain the LDS-DMA ring - before the loop's first MFMA
#pragma unroll
    for (int t = 0; t < QT; ++t) {
#pragma unroll
        for (int s2 = 0; s2 < NMAIN; ++s2) asm volatile("" ::"v"(qm[t][s2]));
        asm volatile("" ::"v"(qr[t]));
    }
#endif
    f32x4 oacc[QT][DTO];
    float mfix[QT];   // the row's offset, exactly the fp16 value that sits in the Q fragment
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        mfix[t] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DTO; ++dt) oacc[t][dt] = fzero4();
    }
    // per-lane read bases inside a stage (bytes)
    const int swz = (c15 >> 3) & 1;
    const int k_main = c15 * PB + ((g ^ swz) << 4);
    const int k_rem = c15 * PB + ((4 * NMAIN + ((g >> 1) ^ swz)) << 4) + 8 * (g & 1);
    const int v_tr = IMG + (4 * g + (c15 >> 2)) * PB + 8 * (c15 & 3);
    // the lane of the last V^T fragment that holds row d (= softmax denominators): fp16 1.0 in all its k-slots
    const uint32_t ones_bits = c15 == (PADROW ? (P.d & 15) : 0) ? 0x3C003C00u : 0u;
    u32x4 ones_w = {ones_bits, ones_bits, ones_bits, ones_bits};
    const half8_t ones8 = __builtin_bit_cast(half8_t, ones_w);   // !PADROW: the constant V^T fragment of the extra tile
    // K side of the offset slot: 1.0 for every key
    half4_t k_one = zero4();
    if (g == 0) k_one[0] = (half_t)1.0f;

    for (int kt = 0; kt < nk; ++kt) {
        // my part of tile kt has landed (the younger tile may still be in flight); behind the barrier everyone's has, and
        // every wave is done reading tile kt - 1, whose stage tile kt + 2 overwrites
        if (kt + 1 < nk) wait_vmcnt_le<IPW>();
        else wait_vmcnt_le<0>();
        raw_barrier();
        if (kt + 2 < nk) issue(kt + 2);
        const char* stg = smem + (kt % NSTG) * STAGE;
        const int kv0 = kt * KV_TILE;

        f32x4 st[QT][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const char* kj = stg + j * 16 * PB;
            const half4_t kr = ld4(reinterpret_cast<const half_t*>(kj + k_rem));
#pragma unroll
            for (int s2 = 0; s2 < NMAIN; ++s2) {
                const half8_t kf = ld8(reinterpret_cast<const half_t*>(kj + k_main + 64 * s2));
#pragma unroll
                for (int t = 0; t < QT; ++t) st[t][j] = mfma16k32(kf, qm[t][s2], s2 == 0 ? fzero4() : st[t][j]);
            }
            const half8_t kf = cat4(kr, k_one);
#pragma unroll
            for (int t = 0; t < QT; ++t) st[t][j] = mfma16k32(kf, cat4(qr[t], qx[t]), NMAIN == 0 ? fzero4() : st[t][j]);
        }
        // V^T fragments of the whole tile: [dt][jp] = keys {32 jp + 4 g + e, 32 jp + 16 + 4 g + e} x head-dim 16 dt + c15
        // (issued now, awaited in front of the first PV MFMA: the first query tile's softmax covers their latency)
        half8_t vf[DT][2];
        static_for<DT>([&](auto dt_) {
            constexpr int dt = decltype(dt_)::value;
            vf[dt][0] = cat4(lds_read_tr4_async<32 * dt>(stg + v_tr), lds_read_tr4_async<32 * dt + 16 * PB>(stg + v_tr));
            vf[dt][1] = cat4(lds_read_tr4_async<32 * dt + 32 * PB>(stg + v_tr), lds_read_tr4_async<32 * dt + 48 * PB>(stg + v_tr));
        });
        const bool tail = kv0 + KV_TILE > P.Nk;
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            if (tail) {   // wave-uniform; live keys of this lane group are 16 j + i < Nk - kv0 - 4 g (constants against one value)
                const int live = opaque(P.Nk - kv0 - 4 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (16 * j + i >= live) st[t][j][i] = -INFINITY;
            }
            // st = s * scale * log2(e) - mfix.  Largest of this lane's 16 scores: only to notice a row outgrowing its offset
            float mx = max3(st[t][0][0], st[t][0][1], st[t][0][2]);
            mx = max3(mx, st[t][0][3], st[t][1][0]);
            mx = max3(mx, st[t][1][1], st[t][1][2]);
            mx = max3(mx, st[t][1][3], st[t][2][0]);
            mx = max3(mx, st[t][2][1], st[t][2][2]);
            mx = max3(mx, st[t][2][3], st[t][3][0]);
            mx = max3(mx, st[t][3][1], st[t][3][2]);
            mx = fmaxf(mx, st[t][3][3]);
            if (kt == 0 || wave_any(mx > kRebase)) {
                // re-base: new offset = the row's current maximum (first tile) or the larger of the two, rounded to the fp16 it
                // is stored as; this tile's scores move by the difference on the VALU, the accumulators (output rows and, in the
                // ones row, the denominators) by 2^-difference
                mx = rows_max(mx);
                // (saturating conversion: beyond +-65504 log2 units - where the reference's own fp16 scores are inf - the offset
                // stops following instead of turning into inf)
                float mnew = (float)to_half(mfix[t] + mx);
                if (kt != 0) mnew = fmaxf(mnew, mfix[t]);
                const float delta = mnew - mfix[t];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) st[t][j][i] -= delta;
                const float alpha = fast_exp2(-delta);
#pragma unroll
                for (int dt = 0; dt < DTO; ++dt) scale_in_place(oacc[t][dt], alpha);
                mfix[t] = mnew;
                if (g == 0) qx[t][0] = (half_t)(-mnew);
            }
            half8_t pf[2];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    const half2_t h2 = pk_rtz(fast_exp2(st[t][j][i]), fast_exp2(st[t][j][i + 1]));
                    pf[j >> 1][4 * (j & 1) + i] = h2[0];
                    pf[j >> 1][4 * (j & 1) + i + 1] = h2[1];
                }
            if (t == 0) {
                lds_tr_wait();
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) lds_tr_use(vf[dt][jp]);
                if (PADROW) {
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
                        u32x4 w = __builtin_bit_cast(u32x4, vf[DT - 1][jp]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) w[i] |= ones_bits;
                        vf[DT - 1][jp] = __builtin_bit_cast(half8_t, w);
                    }
                }
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) oacc[t][dt] = mfma16k32(vf[dt][jp], pf[jp], oacc[t][dt]);
            if (!PADROW) {
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) oacc[t][DTO - 1] = mfma16k32(ones8, pf[jp], oacc[t][DTO - 1]);
            }
        }
    }

#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = q0 + 16 * t + c15;
        // O^T row d (the denominators) sits in accumulator tile DT-1, element 0 of lane group (d % 16) / 4
        // (!PADROW: row 0 of the extra tile)
        const float lt = shfl(oacc[t][DTO - 1][0], (PADROW ? 16 * ((P.d & 15) >> 2) : 0) + c15);
        if (qi >= P.Nq) continue;
        const float inv = 1.0f / lt;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int c = 16 * dt + 4 * g;
            if (c < P.d) {
                half4_t ov;
#pragma unroll
                for (int i = 0; i < 4; ++i) ov[i] = to_half(oacc[t][dt][i] * inv);
                st4(o + (qbase + qi) * ldo + col0 + c, ov);
            }
        }
        if (g == 0 && lse) lse[((size_t)b * P.heads + h) * P.Nq + qi] = (mfix[t] + log2f(lt)) * 0.6931471805599453f;
    }
}

// ---- backward: dQ (and D = rowsum(dO * O)) ---------------------------------------------------
// QT query tiles (16 rows each) per wave: every K / V / K^T fragment read from LDS feeds QT MFMAs, and a block's
// staged tile serves 64*QT query rows.
template <int DT, int QT>
__global__ __launch_bounds__(256, (QT == 1 && DT <= 5) ? 4 : 2) void attn_bwd_dq_kernel(
    AParams P, const half_t* o, int ldo, const half_t* dO, int lddo, const float* lse, float* Dbuf, half_t* dq,
    int lddq) {
    constexpr int RP = DT * 16 + 8;
    constexpr int DP = (DT + 1) / 2;
    MC_DYN_SMEM(smem);
    half_t* Ks = reinterpret_cast<half_t*>(smem);  // [64][RP]
    half_t* Vs = Ks + KV_TILE * RP;                // [64][RP]
    half_t* Kt = Vs + KV_TILE * RP;                // [DT*16][TPAD]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c15 = lane & 15;
    int xb, h, b;
    if (!attn_block(P, (P.Nq + 64 * QT - 1) / (64 * QT), xb, h, b)) return;
    const int col0 = h * P.d;
    const size_t qbase = (size_t)b * P.Nq, kbase = (size_t)(b / P.kv_bdiv) * P.Nk;
    const int q0 = xb * (64 * QT) + wave * (16 * QT);
    const float sl2 = P.scale * 1.4426950408889634f;

    zero_pads<DT>(Ks, Kt, P.d);
    zero_pads<DT>(Vs, nullptr, P.d);

    // K=16 fragment pairs concatenated for the K=32 MFMA (odd last step padded with zeros)
    half8_t qf8[QT][DP], dof8[QT][DP];
    float Dq[QT], lq2[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = q0 + 16 * t + c15;
        const bool qok = qi < P.Nq;
        half4_t qf[2 * DP], dof[2 * DP];
        float dsum = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2 * DP; ++ks) {
            int c = 16 * ks + 4 * g;
            if (qok && c < P.d) {
                qf[ks] = ld4(P.q + (qbase + qi) * P.ldq + col0 + c);
                dof[ks] = ld4(dO + (qbase + qi) * lddo + col0 + c);
                half4_t ov = ld4(o + (qbase + qi) * ldo + col0 + c);
#pragma unroll
                for (int i = 0; i < 4; ++i) dsum += (float)dof[ks][i] * (float)ov[i];
            } else {
                qf[ks] = zero4();
                dof[ks] = zero4();
            }
        }
#pragma unroll
        for (int kp = 0; kp < DP; ++kp) {
            qf8[t][kp] = cat4(qf[2 * kp], qf[2 * kp + 1]);
            dof8[t][kp] = cat4(dof[2 * kp], dof[2 * kp + 1]);
        }
        Dq[t] = grp_sum(dsum);
        const size_t sidx = ((size_t)b * P.heads + h) * P.Nq + (qok ? qi : 0);
        lq2[t] = qok ? lse[sidx] * 1.4426950408889634f : 0.f;   // log-sum-exp in log2 units
        if (qok && g == 0 && Dbuf) Dbuf[sidx] = Dq[t];
    }

    f32x4 acc[QT][DT];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[t][dt] = fzero4();

    for (int kv0 = 0; kv0 < P.Nk; kv0 += KV_TILE) {
        __syncthreads();
        stage_rows<DT>(Ks, P.k, kbase, kv0, P.Nk, P.ldk, col0, P.d);
        stage_cols(Kt, P.k, kbase, kv0, P.Nk, P.ldk, col0, P.d);
        stage_rows<DT>(Vs, P.v, kbase, kv0, P.Nk, P.ldv, col0, P.d);
        __syncthreads();
        const bool tail = kv0 + KV_TILE > P.Nk;
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
            half4_t dsf[QT][2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * jp + jj;
                f32x4 sT[QT], dpT[QT];
                const half_t* krow = Ks + (16 * j + c15) * RP + 4 * g;
                const half_t* vrow = Vs + (16 * j + c15) * RP + 4 * g;
#pragma unroll
                for (int kp = 0; kp < DP; ++kp) {
                    const bool full = 2 * kp + 1 < DT;
                    half8_t kf = cat4(ld4(krow + 32 * kp), full ? ld4(krow + 32 * kp + 16) : zero4());
                    half8_t vf = cat4(ld4(vrow + 32 * kp), full ? ld4(vrow + 32 * kp + 16) : zero4());
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        sT[t] = mfma16k32(kf, qf8[t][kp], kp == 0 ? fzero4() : sT[t]);
                        dpT[t] = mfma16k32(vf, dof8[t][kp], kp == 0 ? fzero4() : dpT[t]);
                    }
                }
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    float ds[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float pr = fast_exp2(fmaf(sT[t][i], sl2, -lq2[t]));
                        if (tail && kv0 + 16 * j + 4 * g + i >= P.Nk) pr = 0.f;
                        ds[i] = pr * (dpT[t][i] - Dq[t]);
                    }
                    half2_t a = pk_rtz(ds[0], ds[1]), c = pk_rtz(ds[2], ds[3]);
                    dsf[t][jj][0] = a[0]; dsf[t][jj][1] = a[1]; dsf[t][jj][2] = c[0]; dsf[t][jj][3] = c[1];
                }
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const half_t* kc = Kt + (16 * dt + c15) * TPAD + 32 * jp + 4 * g;
                const half8_t k8 = cat4(ld4(kc), ld4(kc + 16));
#pragma unroll
                for (int t = 0; t < QT; ++t) acc[t][dt] = mfma16k32(k8, cat4(dsf[t][0], dsf[t][1]), acc[t][dt]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = q0 + 16 * t + c15;
        if (qi >= P.Nq) continue;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            int c = 16 * dt + 4 * g;
            if (c < P.d) {
                half4_t ov;
#pragma unroll
                for (int i = 0; i < 4; ++i) ov[i] = to_half(acc[t][dt][i] * P.scale);
                st4(dq + (qbase + qi) * lddq + col0 + c, ov);
            }
        }
    }
}

// ---- backward: dK, dV (self-attention: kv batch == q batch) -----------------------------------
// KT key/value tiles (16 rows each) per wave, same sharing of the staged Q / dO / Q^T / dO^T fragments.
template <int DT, int KT>
__global__ __launch_bounds__(256, (KT == 1 && DT <= 4) ? 4 : ((KT == 1 && DT == 5) ? 3 : 2)) void attn_bwd_dkdv_kernel(
    AParams P, const half_t* dO, int lddo, const float* lse, const float* Dbuf, half_t* dk, int lddk, half_t* dv,
    int lddv) {
    constexpr int RP = DT * 16 + 8;
    constexpr int DP = (DT + 1) / 2;
    MC_DYN_SMEM(smem);
    half_t* Qs = reinterpret_cast<half_t*>(smem);  // [64][RP]
    half_t* Os = Qs + KV_TILE * RP;                // [64][RP]   dO rows
    half_t* Qt = Os + KV_TILE * RP;                // [DT*16][TPAD]
    half_t* Ot = Qt + DT * 16 * TPAD;              // [DT*16][TPAD]
    float* lse_s = reinterpret_cast<float*>(Ot + DT * 16 * TPAD);  // [64]
    float* D_s = lse_s + KV_TILE;                                   // [64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c15 = lane & 15;
    int xb, h, b;
    if (!attn_block(P, (P.Nk + 64 * KT - 1) / (64 * KT), xb, h, b)) return;
    const int col0 = h * P.d;
    const size_t qbase = (size_t)b * P.Nq, kbase = (size_t)b * P.Nk;
    const int k0 = xb * (64 * KT) + wave * (16 * KT);
    const float sl2 = P.scale * 1.4426950408889634f;

    zero_pads<DT>(Qs, Qt, P.d);
    zero_pads<DT>(Os, Ot, P.d);

    half8_t kf8[KT][DP], vf8[KT][DP];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const int kvi = k0 + 16 * t + c15;
        half4_t kf[2 * DP], vf[2 * DP];
#pragma unroll
        for (int ks = 0; ks < 2 * DP; ++ks) {
            int c = 16 * ks + 4 * g;
            if (kvi < P.Nk && c < P.d) {
                kf[ks] = ld4(P.k + (kbase + kvi) * P.ldk + col0 + c);
                vf[ks] = ld4(P.v + (kbase + kvi) * P.ldv + col0 + c);
            } else {
                kf[ks] = zero4();
                vf[ks] = zero4();
            }
        }
#pragma unroll
        for (int kp = 0; kp < DP; ++kp) {
            kf8[t][kp] = cat4(kf[2 * kp], kf[2 * kp + 1]);
            vf8[t][kp] = cat4(vf[2 * kp], vf[2 * kp + 1]);
        }
    }
    f32x4 ak[KT][DT], av[KT][DT];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) ak[t][dt] = av[t][dt] = fzero4();

    for (int q0 = 0; q0 < P.Nq; q0 += KV_TILE) {
        __syncthreads();
        stage_rows<DT>(Qs, P.q, qbase, q0, P.Nq, P.ldq, col0, P.d);
        stage_cols(Qt, P.q, qbase, q0, P.Nq, P.ldq, col0, P.d);
        stage_rows<DT>(Os, dO, qbase, q0, P.Nq, lddo, col0, P.d);
        stage_cols(Ot, dO, qbase, q0, P.Nq, lddo, col0, P.d);
        if (threadIdx.x < KV_TILE) {
            int qi = q0 + threadIdx.x;
            size_t si = ((size_t)b * P.heads + h) * P.Nq + qi;
            lse_s[threadIdx.x] = qi < P.Nq ? lse[si] * 1.4426950408889634f : INFINITY;   // log2 units
            D_s[threadIdx.x] = qi < P.Nq ? Dbuf[si] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
            half4_t pf[KT][2], dsf[KT][2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * jp + jj;
                f32x4 sc[KT], dp[KT];
                const half_t* qrow = Qs + (16 * j + c15) * RP + 4 * g;
                const half_t* orow = Os + (16 * j + c15) * RP + 4 * g;
#pragma unroll
                for (int kp = 0; kp < DP; ++kp) {
                    const bool full = 2 * kp + 1 < DT;
                    half8_t qr = cat4(ld4(qrow + 32 * kp), full ? ld4(qrow + 32 * kp + 16) : zero4());
                    half8_t orr = cat4(ld4(orow + 32 * kp), full ? ld4(orow + 32 * kp + 16) : zero4());
#pragma unroll
                    for (int t = 0; t < KT; ++t) {
                        sc[t] = mfma16k32(qr, kf8[t][kp], kp == 0 ? fzero4() : sc[t]);   // [q = 16j + 4g + i][kv = c15]
                        dp[t] = mfma16k32(orr, vf8[t][kp], kp == 0 ? fzero4() : dp[t]);
                    }
                }
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + 16 * j + 4 * g);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(D_s + 16 * j + 4 * g);
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    float pr[4], ds[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        pr[i] = fast_exp2(fmaf(sc[t][i], sl2, -l4[i]));
                        ds[i] = pr[i] * (dp[t][i] - d4[i]);
                    }
                    half2_t a = pk_rtz(pr[0], pr[1]), c = pk_rtz(pr[2], pr[3]);
                    pf[t][jj][0] = a[0]; pf[t][jj][1] = a[1]; pf[t][jj][2] = c[0]; pf[t][jj][3] = c[1];
                    a = pk_rtz(ds[0], ds[1]); c = pk_rtz(ds[2], ds[3]);
                    dsf[t][jj][0] = a[0]; dsf[t][jj][1] = a[1]; dsf[t][jj][2] = c[0]; dsf[t][jj][3] = c[1];
                }
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const half_t* oc = Ot + (16 * dt + c15) * TPAD + 32 * jp + 4 * g;
                const half_t* qc = Qt + (16 * dt + c15) * TPAD + 32 * jp + 4 * g;
                const half8_t o8 = cat4(ld4(oc), ld4(oc + 16)), q8 = cat4(ld4(qc), ld4(qc + 16));
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    av[t][dt] = mfma16k32(o8, cat4(pf[t][0], pf[t][1]), av[t][dt]);
                    ak[t][dt] = mfma16k32(q8, cat4(dsf[t][0], dsf[t][1]), ak[t][dt]);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const int kvi = k0 + 16 * t + c15;
        if (kvi >= P.Nk) continue;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            int c = 16 * dt + 4 * g;
            if (c < P.d) {
                half4_t ok, ov;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ok[i] = to_half(ak[t][dt][i] * P.scale);
                    ov[i] = to_half(av[t][dt][i]);
                }
                st4(dk + (kbase + kvi) * lddk + col0 + c, ok);
                st4(dv + (kbase + kvi) * lddv + col0 + c, ov);
            }
        }
    }
}

// ---- backward at d = 40, long sequences: the same LDS-DMA ring around the backward MFMAs --------------------------------------
// Shared by the two kernels below: a 64-row tile of a [rows, ld] fp16 matrix (this head's d columns) as a row-major LDS image
// [64][32 DT bytes]; chunks at / beyond d are fetched out of range (hardware zeros); the 16-byte chunks of rows 8..15 of every
// 16 are swapped pairwise (conflict-free 16-byte row reads).  Both operand orientations are read from this one image: rows as
// MFMA fragments directly, columns through the transpose read - the transposed LDS copies of the old kernels are gone.
template <int DT>
struct RingImage {
    static constexpr int CPR = 2 * DT, PB = 32 * DT, BYTES = KV_TILE * PB;
    // byte offset (from the buffer base) of what LDS-DMA instruction `inst` (0 .. CPR-1) of this image fetches for `lane`
    static __device__ __forceinline__ uint32_t dma_offset(int inst, int lane, int ld, int col0, int vpr) {
        const int pch = inst * 64 + lane;
        const int row = pch / CPR, cp = pch % CPR;
        const int c = cp ^ ((row >> 3) & 1);
        return c < vpr ? (uint32_t)(row * ld + col0 + 8 * c) * 2 : kOOB;
    }
    // row fragments of row 16 j + c15: 32-wide step s (head-dim [32 s + 8 g, +8)) and the 16-wide remainder ([32 NMAIN + 4 g, +4))
    static __device__ __forceinline__ int row_main(int c15, int g) { return c15 * PB + ((g ^ ((c15 >> 3) & 1)) << 4); }
    static __device__ __forceinline__ int row_rem(int c15, int g) {
        return c15 * PB + ((4 * (DT / 2) + ((g >> 1) ^ ((c15 >> 3) & 1))) << 4) + 8 * (g & 1);
    }
    // transpose-read base of a lane: rows {r0 + 4 g + e}, column 16 dt + c15 -> + (r0 * PB + 32 dt) as the immediate
    static __device__ __forceinline__ int col_base(int c15, int g) {
        return (4 * g + (c15 >> 2)) * PB + (((((c15 & 3) >> 1) ^ ((g >> 1) & 1))) << 4) + 8 * (c15 & 1);
    }
};

// dQ (and D = rowsum(dO * O)).  S^T and dP^T tiles as in attn_bwd_dq_kernel, with BOTH per-row constants inside the MFMAs: Q is
// pre-multiplied by scale * log2(e) and two padding k-slots carry -lse[q] * log2(e) (hi + lo fp16 parts: this one is the real
// normaliser, its rounding does not cancel) against 1.0 on the K side, so P = exp2(accumulator); two more carry -D[q] against
// 1.0 on the V side, so the second accumulator is dP - D.  Per score: one v_exp_f32, one multiply, half a packed convert.
// Keys past Nk need no mask: their K rows are hardware zeros, so whatever dS they produce multiplies a zero row in dQ += dS K.
template <int DT, int QT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_ring_kernel(AParams P, const half_t* o, int ldo, const half_t* dO,
                                                                   int lddo, const float* lse, float* Dbuf, half_t* dq, int lddq) {
    using Img = RingImage<DT>;
    constexpr int PB = Img::PB, IMG = Img::BYTES, STAGE = 2 * IMG, NSTG = 3, CPR = Img::CPR;
    constexpr int IPW = 2 * CPR / 4;
    constexpr int NMAIN = DT / 2;
    static_assert(DT & 1, "needs the padding k-slots of the 16-wide remainder step");
    MC_DYN_SMEM(smem);
    const int lane = threadIdx.x & 63;
#ifdef MC_EMU
    const int wave = threadIdx.x >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#endif
    const int g = lane >> 4, c15 = lane & 15;
    int xb, h, b;
    if (!attn_block(P, (P.Nq + 64 * QT - 1) / (64 * QT), xb, h, b)) return;
    const int col0 = h * P.d;
    const size_t qbase = (size_t)b * P.Nq, kbase = (size_t)(b / P.kv_bdiv) * P.Nk;
    const int q0 = xb * (64 * QT) + wave * (16 * QT);
    const int vpr = P.d / 8;
    const float sl2 = P.scale * 1.4426950408889634f;
    const int nk = (P.Nk + KV_TILE - 1) / KV_TILE;

    const GBuf kbuf = make_gbuf(P.k + kbase * P.ldk, (uint32_t)P.Nk * P.ldk * 2);
    const GBuf vbuf = make_gbuf(P.v + kbase * P.ldv, (uint32_t)P.Nk * P.ldv * 2);
    uint32_t dma_off[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int n = wave + 4 * i;
        dma_off[i] = n >= CPR ? Img::dma_offset(n - CPR, lane, P.ldv, col0, vpr) : Img::dma_offset(n, lane, P.ldk, col0, vpr);
    }
    auto issue = [&](int tile) {
        char* stg = smem + (tile % NSTG) * STAGE;
        const uint32_t ko = (uint32_t)tile * KV_TILE * P.ldk * 2, vo = (uint32_t)tile * KV_TILE * P.ldv * 2;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int n = wave + 4 * i;
            if (n >= CPR) glds16(vbuf, dma_off[i] + vo, stg + IMG + (n - CPR) * 1024);
            else glds16(kbuf, dma_off[i] + ko, stg + n * 1024);
        }
    };

    // Q' = Q * scale * log2(e) and dO as B-operand fragments; the remainder step's upper half holds the row constants
    half8_t qm[QT][NMAIN > 0 ? NMAIN : 1], dom[QT][NMAIN > 0 ? NMAIN : 1];
    half4_t qr[QT], qx[QT], dor[QT], dx[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = q0 + 16 * t + c15;
        const bool qok = qi < P.Nq;
        const half_t* qp = P.q + (qbase + qi) * P.ldq + col0;
        const half_t* dp = dO + (qbase + qi) * lddo + col0;
        const half_t* op = o + (qbase + qi) * ldo + col0;
        float dsum = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < NMAIN; ++s2) {
            const half8_t qv = qok ? ld8(qp + 32 * s2 + 8 * g) : zero8();
            dom[t][s2] = qok ? ld8(dp + 32 * s2 + 8 * g) : zero8();
            const half8_t ov = qok ? ld8(op + 32 * s2 + 8 * g) : zero8();
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                qm[t][s2][e] = (half_t)((float)qv[e] * sl2);
                dsum += (float)dom[t][s2][e] * (float)ov[e];
            }
        }
        const bool rok = qok && 32 * NMAIN + 4 * g < P.d;
        const half4_t qv = rok ? ld4(qp + 32 * NMAIN + 4 * g) : zero4();
        dor[t] = rok ? ld4(dp + 32 * NMAIN + 4 * g) : zero4();
        const half4_t ov = rok ? ld4(op + 32 * NMAIN + 4 * g) : zero4();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            qr[t][e] = (half_t)((float)qv[e] * sl2);
            dsum += (float)dor[t][e] * (float)ov[e];
        }
        const float Dq = rows_sum(dsum);
        const size_t sidx = ((size_t)b * P.heads + h) * P.Nq + (qok ? qi : 0);
        const float lq2 = qok ? lse[sidx] * 1.4426950408889634f : 0.f;
        if (qok && g == 0 && Dbuf) Dbuf[sidx] = Dq;
        qx[t] = zero4();
        dx[t] = zero4();
        if (g == 0) {
            const half_t lh = (half_t)lq2, dh = (half_t)Dq;
            qx[t][0] = -lh;
            qx[t][1] = (half_t)((float)lh - lq2);
            dx[t][0] = -dh;
            dx[t][1] = (half_t)((float)dh - Dq);
        }
    }
    issue(0);
    if (nk > 1) issue(1);
#ifndef MC_EMU
#pragma unroll
    for (int t = 0; t < QT; ++t) {   // pins hipcc's wait for the loads above in front of the loop (see attn_fwd_ring_kernel)
#pragma unroll
        for (int s2 = 0; s2 < NMAIN; ++s2) {
            asm volatile("" ::"v"(qm[t][s2]));
            asm volatile("" ::"v"(dom[t][s2]));
        }
        asm volatile("" ::"v"(qr[t]), "v"(dor[t]), "v"(qx[t]), "v"(dx[t]));
    }
#endif
    f32x4 acc[QT][DT];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[t][dt] = fzero4();
    const int r_main = Img::row_main(c15, g), r_rem = Img::row_rem(c15, g), c_base = Img::col_base(c15, g);
    half4_t one2 = zero4();   // 1.0 in the two constant slots, K / V side
    if (g == 0) one2[0] = one2[1] = (half_t)1.0f;

    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) wait_vmcnt_le<IPW>();
        else wait_vmcnt_le<0>();
        raw_barrier();
        if (kt + 2 < nk) issue(kt + 2);
        const char* stg = smem + (kt % NSTG) * STAGE;
        static_for<2>([&](auto jp_) {
            constexpr int jp = decltype(jp_)::value;
            // K^T fragments of this half tile (keys 32 jp ..): issued now, awaited in front of the dQ MFMAs
            half8_t k8[DT];
            static_for<DT>([&](auto dt_) {
                constexpr int dt = decltype(dt_)::value;
                k8[dt] = cat4(lds_read_tr4_async<32 * jp * PB + 32 * dt>(stg + c_base),
                              lds_read_tr4_async<(32 * jp + 16) * PB + 32 * dt>(stg + c_base));
            });
            half4_t dsf[QT][2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const char* kj = stg + (2 * jp + jj) * 16 * PB;
                f32x4 sT[QT], dpT[QT];
#pragma unroll
                for (int s2 = 0; s2 < NMAIN; ++s2) {
                    const half8_t kf = ld8(reinterpret_cast<const half_t*>(kj + r_main + 64 * s2));
                    const half8_t vf = ld8(reinterpret_cast<const half_t*>(kj + IMG + r_main + 64 * s2));
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        sT[t] = mfma16k32(kf, qm[t][s2], s2 == 0 ? fzero4() : sT[t]);
                        dpT[t] = mfma16k32(vf, dom[t][s2], s2 == 0 ? fzero4() : dpT[t]);
                    }
                }
                const half8_t kf = cat4(ld4(reinterpret_cast<const half_t*>(kj + r_rem)), one2);
                const half8_t vf = cat4(ld4(reinterpret_cast<const half_t*>(kj + IMG + r_rem)), one2);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    sT[t] = mfma16k32(kf, cat4(qr[t], qx[t]), NMAIN == 0 ? fzero4() : sT[t]);
                    dpT[t] = mfma16k32(vf, cat4(dor[t], dx[t]), NMAIN == 0 ? fzero4() : dpT[t]);
                }
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    const half2_t a = pk_rtz(fast_exp2(sT[t][0]) * dpT[t][0], fast_exp2(sT[t][1]) * dpT[t][1]);
                    const half2_t c = pk_rtz(fast_exp2(sT[t][2]) * dpT[t][2], fast_exp2(sT[t][3]) * dpT[t][3]);
                    dsf[t][jj][0] = a[0]; dsf[t][jj][1] = a[1]; dsf[t][jj][2] = c[0]; dsf[t][jj][3] = c[1];
                }
            }
            lds_tr_wait();
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) lds_tr_use(k8[dt]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int t = 0; t < QT; ++t) acc[t][dt] = mfma16k32(k8[dt], cat4(dsf[t][0], dsf[t][1]), acc[t][dt]);
        });
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = q0 + 16 * t + c15;
        if (qi >= P.Nq) continue;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int c = 16 * dt + 4 * g;
            if (c < P.d) {
                half4_t ov;
#pragma unroll
                for (int i = 0; i < 4; ++i) ov[i] = to_half(acc[t][dt][i] * P.scale);
                st4(dq + (qbase + qi) * lddq + col0 + c, ov);
            }
        }
    }
}

// dK, dV (self-attention).  Q and dO tiles (64 query rows) stream through the ring together with the rows' lse and D (one more
// 16-byte-per-lane LDS-DMA instruction per tile: 64 + 64 floats); K (pre-multiplied by scale * log2(e)) and V stay in registers as
// B-operand fragments.  The row constants vary along the ACCUMULATOR ROWS here (A side = the LDS images), so they stay on the
// VALU: per score one subtract, one v_exp_f32, a subtract and a multiply, one packed convert.  Query rows past Nq are hardware
// zeros with lse = D = 0: P = 1 and dO = 0 there, so they add nothing to dV or dK.
template <int DT, int KT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_ring_kernel(AParams P, const half_t* dO, int lddo, const float* lse,
                                                                     const float* Dbuf, half_t* dk, int lddk, half_t* dv, int lddv) {
    using Img = RingImage<DT>;
    constexpr int PB = Img::PB, IMG = Img::BYTES, STAGE = 2 * IMG + 2048, NSTG = 3, CPR = Img::CPR;
    constexpr int IPW = 2 * CPR / 4;     // + 1 on waves 0 and 1: the lse / D chunks
    constexpr int NMAIN = DT / 2;
    static_assert(DT & 1, "written for the 16-wide remainder step");
    MC_DYN_SMEM(smem);
    const int lane = threadIdx.x & 63;
#ifdef MC_EMU
    const int wave = threadIdx.x >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#endif
    const int g = lane >> 4, c15 = lane & 15;
    int xb, h, b;
    if (!attn_block(P, (P.Nk + 64 * KT - 1) / (64 * KT), xb, h, b)) return;
    const int col0 = h * P.d;
    const size_t qbase = (size_t)b * P.Nq, kbase = (size_t)b * P.Nk;
    const int k0 = xb * (64 * KT) + wave * (16 * KT);
    const int vpr = P.d / 8;
    const float sl2 = P.scale * 1.4426950408889634f;
    const int nq = (P.Nq + KV_TILE - 1) / KV_TILE;

    const GBuf qbuf = make_gbuf(P.q + qbase * P.ldq, (uint32_t)P.Nq * P.ldq * 2);
    const GBuf obuf = make_gbuf(dO + qbase * lddo, (uint32_t)P.Nq * lddo * 2);
    const size_t stat0 = ((size_t)b * P.heads + h) * P.Nq;
    const GBuf lbuf = make_gbuf(lse + stat0, (uint32_t)P.Nq * 4);
    const GBuf dbuf = make_gbuf(Dbuf + stat0, (uint32_t)P.Nq * 4);
    uint32_t dma_off[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int n = wave + 4 * i;
        dma_off[i] = n >= CPR ? Img::dma_offset(n - CPR, lane, lddo, col0, vpr) : Img::dma_offset(n, lane, P.ldq, col0, vpr);
    }
    // lanes 0..15 of one instruction: the tile's 64 lse (wave 0) or D (wave 1) values; the other lanes write zeros behind them
    const uint32_t stat_off = lane < 16 ? (uint32_t)lane * 16 : kOOB;
    auto issue = [&](int tile) {
        char* stg = smem + (tile % NSTG) * STAGE;
        const uint32_t qo = (uint32_t)tile * KV_TILE * P.ldq * 2, oo = (uint32_t)tile * KV_TILE * lddo * 2;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int n = wave + 4 * i;
            if (n >= CPR) glds16(obuf, dma_off[i] + oo, stg + IMG + (n - CPR) * 1024);
            else glds16(qbuf, dma_off[i] + qo, stg + n * 1024);
        }
        const uint32_t so = stat_off + (uint32_t)tile * KV_TILE * 4;
        if (wave == 0) glds16(lbuf, so, stg + 2 * IMG);
        if (wave == 1) glds16(dbuf, so, stg + 2 * IMG + 1024);
    };

    // K' = K * scale * log2(e) and V as B-operand fragments (remainder step zero-extended)
    // (remainder step: K in k-slots 0-3 and V in k-slots 4-7 of ONE fragment; the Q / dO side zeroes the half it does not want)
    half8_t km[KT][NMAIN > 0 ? NMAIN : 1], vm[KT][NMAIN > 0 ? NMAIN : 1], kvr[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const int kvi = k0 + 16 * t + c15;
        const bool ok = kvi < P.Nk;
        const half_t* kp = P.k + (kbase + kvi) * P.ldk + col0;
        const half_t* vp = P.v + (kbase + kvi) * P.ldv + col0;
#pragma unroll
        for (int s2 = 0; s2 < NMAIN; ++s2) {
            const half8_t kv = ok ? ld8(kp + 32 * s2 + 8 * g) : zero8();
            vm[t][s2] = ok ? ld8(vp + 32 * s2 + 8 * g) : zero8();
#pragma unroll
            for (int e = 0; e < 8; ++e) km[t][s2][e] = (half_t)((float)kv[e] * sl2);
        }
        const bool rok = ok && 32 * NMAIN + 4 * g < P.d;
        half4_t kv = rok ? ld4(kp + 32 * NMAIN + 4 * g) : zero4();
#pragma unroll
        for (int e = 0; e < 4; ++e) kv[e] = (half_t)((float)kv[e] * sl2);
        kvr[t] = cat4(kv, rok ? ld4(vp + 32 * NMAIN + 4 * g) : zero4());
    }
    issue(0);
    if (nq > 1) issue(1);
#ifndef MC_EMU
#pragma unroll
    for (int t = 0; t < KT; ++t) {
#pragma unroll
        for (int s2 = 0; s2 < NMAIN; ++s2) {
            asm volatile("" ::"v"(km[t][s2]));
            asm volatile("" ::"v"(vm[t][s2]));
        }
        asm volatile("" ::"v"(kvr[t]));
    }
#endif
    f32x4 ak[KT][DT], av[KT][DT];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) ak[t][dt] = av[t][dt] = fzero4();
    const int r_main = Img::row_main(c15, g), r_rem = Img::row_rem(c15, g), c_base = Img::col_base(c15, g);

    for (int qt = 0; qt < nq; ++qt) {
        // waves 0 and 1 have one more instruction per tile in flight (the lse / D chunks)
        if (qt + 1 < nq) {
            if (wave < 2) wait_vmcnt_le<IPW + 1>();
            else wait_vmcnt_le<IPW>();
        } else {
            wait_vmcnt_le<0>();
        }
        raw_barrier();
        if (qt + 2 < nq) issue(qt + 2);
        const char* stg = smem + (qt % NSTG) * STAGE;
        const float* stat = reinterpret_cast<const float*>(stg + 2 * IMG);   // [64] lse, +256 floats: [64] D
        static_for<2>([&](auto jp_) {
            constexpr int jp = decltype(jp_)::value;
            // dO^T and Q^T fragments of this half tile (query rows 32 jp ..): read one head-dim tile ahead of their MFMAs
            half8_t o8[2], q8[2];
            auto fetch = [&](auto dt_, int slot) {
                constexpr int dt = decltype(dt_)::value;
                q8[slot] = cat4(lds_read_tr4_async<32 * jp * PB + 32 * dt>(stg + c_base),
                                lds_read_tr4_async<(32 * jp + 16) * PB + 32 * dt>(stg + c_base));
                o8[slot] = cat4(lds_read_tr4_async<IMG + 32 * jp * PB + 32 * dt>(stg + c_base),
                                lds_read_tr4_async<IMG + (32 * jp + 16) * PB + 32 * dt>(stg + c_base));
            };
            half4_t pf[KT][2], dsf[KT][2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * jp + jj;
                const char* qj = stg + j * 16 * PB;
                f32x4 sc[KT], dp[KT];
#pragma unroll
                for (int s2 = 0; s2 < NMAIN; ++s2) {
                    const half8_t qf = ld8(reinterpret_cast<const half_t*>(qj + r_main + 64 * s2));
                    const half8_t of = ld8(reinterpret_cast<const half_t*>(qj + IMG + r_main + 64 * s2));
#pragma unroll
                    for (int t = 0; t < KT; ++t) {
                        sc[t] = mfma16k32(qf, km[t][s2], s2 == 0 ? fzero4() : sc[t]);   // [q = 16 j + 4 g + i][kv = c15]
                        dp[t] = mfma16k32(of, vm[t][s2], s2 == 0 ? fzero4() : dp[t]);
                    }
                }
                const half8_t qf = cat4(ld4(reinterpret_cast<const half_t*>(qj + r_rem)), zero4());
                const half8_t of = cat4(zero4(), ld4(reinterpret_cast<const half_t*>(qj + IMG + r_rem)));
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    sc[t] = mfma16k32(qf, kvr[t], NMAIN == 0 ? fzero4() : sc[t]);
                    dp[t] = mfma16k32(of, kvr[t], NMAIN == 0 ? fzero4() : dp[t]);
                }
                f32x4 l4 = *reinterpret_cast<const f32x4*>(stat + 16 * j + 4 * g);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(stat + 256 + 16 * j + 4 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i) l4[i] *= 1.4426950408889634f;
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    float pr[4], ds[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        pr[i] = fast_exp2(sc[t][i] - l4[i]);
                        ds[i] = pr[i] * (dp[t][i] - d4[i]);
                    }
                    half2_t a = pk_rtz(pr[0], pr[1]), c = pk_rtz(pr[2], pr[3]);
                    pf[t][jj][0] = a[0]; pf[t][jj][1] = a[1]; pf[t][jj][2] = c[0]; pf[t][jj][3] = c[1];
                    a = pk_rtz(ds[0], ds[1]); c = pk_rtz(ds[2], ds[3]);
                    dsf[t][jj][0] = a[0]; dsf[t][jj][1] = a[1]; dsf[t][jj][2] = c[0]; dsf[t][jj][3] = c[1];
                }
            }
            fetch(std::integral_constant<int, 0>{}, 0);
            static_for<DT>([&](auto dt_) {
                constexpr int dt = decltype(dt_)::value;
                lds_tr_wait();
                lds_tr_use(o8[dt & 1]);
                lds_tr_use(q8[dt & 1]);
                if constexpr (dt + 1 < DT) fetch(std::integral_constant<int, dt + 1>{}, (dt + 1) & 1);
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    av[t][dt] = mfma16k32(o8[dt & 1], cat4(pf[t][0], pf[t][1]), av[t][dt]);
                    ak[t][dt] = mfma16k32(q8[dt & 1], cat4(dsf[t][0], dsf[t][1]), ak[t][dt]);
                }
            });
        });
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const int kvi = k0 + 16 * t + c15;
        if (kvi >= P.Nk) continue;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int c = 16 * dt + 4 * g;
            if (c < P.d) {
                half4_t ok, ov;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ok[i] = to_half(ak[t][dt][i] * P.scale);
                    ov[i] = to_half(av[t][dt][i]);
                }
                st4(dk + (kbase + kvi) * lddk + col0 + c, ok);
                st4(dv + (kbase + kvi) * lddv + col0 + c, ov);
            }
        }
    }
}

static int a_check(const AParams& P) {
    if (P.Nq <= 0 || P.Nk <= 0 || P.heads <= 0 || P.d <= 0 || P.nbatch <= 0 || P.kv_bdiv <= 0) return 0;
    if (P.d % 8 || P.ldq % 8 || P.ldk % 8 || P.ldv % 8) return 0;
    return 1;
}

template <int DT, int QT, bool PF>
static void a_launch_fwd_cfg(const AParams& P, half_t* o, int ldo, float* lse, hipStream_t s) {
    constexpr int RP = DT * 16 + 8;
    size_t smem = (size_t)(PF ? 2 : 1) * (KV_TILE * RP + DT * 16 * TPAD) * sizeof(half_t);
    allow_big_smem(attn_fwd_kernel<DT, QT, PF>, smem);
    dim3 grid = attn_grid(P, (P.Nq + 64 * QT - 1) / (64 * QT));
    MC_LAUNCH((attn_fwd_kernel<DT, QT, PF>), grid, dim3(256), smem, s, P, o, ldo, lse);
}

// (query tiles per wave, register prefetch of the next K/V tile), chosen by measurement on MI355X
// (profiles/r01_attention_config_sweep.txt): two query tiles per wave win at every level (half the LDS traffic per
// query); the register prefetch never pays - the loop is VALU-bound on the softmax and occupancy matters more.
// MC_ATTN_QT / MC_ATTN_PF override the choice for experiments.
template <int DT, int QT, bool PADROW>
static void a_launch_fwd_ring(const AParams& P, half_t* o, int ldo, float* lse, hipStream_t s) {
    size_t smem = (size_t)3 * 2 * KV_TILE * 32 * DT;
    allow_big_smem(attn_fwd_ring_kernel<DT, QT, PADROW>, smem);
    dim3 grid = attn_grid(P, (P.Nq + 64 * QT - 1) / (64 * QT));
    MC_LAUNCH((attn_fwd_ring_kernel<DT, QT, PADROW>), grid, dim3(256), smem, s, P, o, ldo, lse);
}

// which structure the calling thread's last spatial-attention entry ran (profiling / tests only): 0 register-staged kernels,
// 1 LDS-DMA ring kernels; for mc_attn_bwd_f16 bit 0 = dQ kernel, bit 1 = dK/dV kernel
static thread_local int g_attn_last = 0;

// MC_ATTN_RING: 0 = never the LDS-DMA ring kernel, 2 = at every size it supports (tests), default = long sequences
static int attn_ring_env() {
    return MC_ENV_INT("MC_ATTN_RING", 1);   // simulator / tools builds: re-read per call (the tests flip it); product: constant 1
}

template <int DT>
static void a_launch_fwd(const AParams& P, half_t* o, int ldo, float* lse, hipStream_t s) {
    static const int qt_env = MC_ENV_INT("MC_ATTN_QT", 0);
    static const int pf_env = MC_ENV_INT("MC_ATTN_PF", -1);
    if constexpr (DT == 3 || DT == 5) {
        const int ring = attn_ring_env();
        // d = 40 / 80 exactly: the ring kernels need the padding k-slots of the 16-wide remainder step (and say which row of
        // V^T carries the denominators)
        // (any number of keys: the 77-key cross-attention gains 28 % as well - it is a stream over Q and O)
        if (!P.causal && P.d == (DT == 3 ? 40 : 80) && (ring == 2 || (ring && P.Nq >= 1024))) {
            g_attn_last = 1;
            if constexpr (DT == 3) {
                // MC_ATTN_QT=2: 32 rows per wave, 122 registers, four workgroups per CU - A/B only: 3-9 % SLOWER than 64 rows per wave
                // at two workgroups per CU (profiles/r03_attn_qt_ab.jsonl): twice the LDS fragment reads and DMA issue per score
                if (qt_env == 2) a_launch_fwd_ring<3, 2, true>(P, o, ldo, lse, s);
                else a_launch_fwd_ring<3, 4, true>(P, o, ldo, lse, s);
            } else {
                a_launch_fwd_ring<5, 2, false>(P, o, ldo, lse, s);
            }
            return;
        }
    }
    bool two = qt_env ? qt_env == 2 : P.Nq >= 256;
    bool pf = pf_env >= 0 ? pf_env != 0 : false;
    if constexpr (DT == 3) {
        // four query tiles per wave (a quarter of the K/V fragment reads per query, 253 registers, 2 waves per SIMD): the
        // 4096-token level of a 512^2 video runs 1.72 -> 1.65 ms at B = 2 (tools/attn_ablate.py)
        if (qt_env == 4 || (!qt_env && P.Nq >= 2048)) {
            a_launch_fwd_cfg<DT, 4, false>(P, o, ldo, lse, s);
            return;
        }
    }
    if (two) {
        if (pf) a_launch_fwd_cfg<DT, 2, true>(P, o, ldo, lse, s);
        else a_launch_fwd_cfg<DT, 2, false>(P, o, ldo, lse, s);
    } else {
        if (pf) a_launch_fwd_cfg<DT, 1, true>(P, o, ldo, lse, s);
        else a_launch_fwd_cfg<DT, 1, false>(P, o, ldo, lse, s);
    }
}
template <int DT, int QT>
static void a_launch_dq_cfg(const AParams& P, const half_t* o, int ldo, const half_t* dO, int lddo, const float* lse,
                            float* Dbuf, half_t* dq, int lddq, hipStream_t s) {
    constexpr int RP = DT * 16 + 8;
    size_t smem = (size_t)(2 * KV_TILE * RP + DT * 16 * TPAD) * sizeof(half_t);
    dim3 grid = attn_grid(P, (P.Nq + 64 * QT - 1) / (64 * QT));
    allow_big_smem(attn_bwd_dq_kernel<DT, QT>, smem);
    MC_LAUNCH((attn_bwd_dq_kernel<DT, QT>), grid, dim3(256), smem, s, P, o, ldo, dO, lddo, lse, Dbuf, dq, lddq);
}
template <int DT, int KT>
static void a_launch_dkdv_cfg(const AParams& P, const half_t* dO, int lddo, const float* lse, const float* Dbuf,
                              half_t* dk, int lddk, half_t* dv, int lddv, hipStream_t s) {
    constexpr int RP = DT * 16 + 8;
    size_t smem = (size_t)(2 * KV_TILE * RP + 2 * DT * 16 * TPAD) * sizeof(half_t) + 2 * KV_TILE * sizeof(float);
    dim3 grid = attn_grid(P, (P.Nk + 64 * KT - 1) / (64 * KT));
    allow_big_smem(attn_bwd_dkdv_kernel<DT, KT>, smem);
    MC_LAUNCH((attn_bwd_dkdv_kernel<DT, KT>), grid, dim3(256), smem, s, P, dO, lddo, lse, Dbuf, dk, lddk, dv, lddv);
}
// two row tiles per wave once there are enough rows to fill the chip with the larger blocks, four for the 4096-token level
// at d = 40 (level-0 backward of a 512^2 video 2.86 -> 2.55 ms, tools/attn_ablate.py; MC_ATTN_BQT overrides);
// head dim 160 keeps one tile (the accumulators alone would need > 256 registers)
static int bwd_tiles(int rows, int dt) {
    static const int env = MC_ENV_INT("MC_ATTN_BQT", 0);
    if (dt > 5) return 1;
    if (env) return env;
    if (dt == 3 && rows >= 2048) return 4;
    return rows >= 512 ? 2 : 1;
}
static bool attn_bwd_ring_wanted(const AParams& P) {
    const int ring = attn_ring_env();
    // d = 40 / 80 exactly (padding k-slots); the D / lse chunks are fetched 16 bytes at a time
    return (P.d == 40 || P.d == 80) && P.Nq % 4 == 0 && (ring == 2 || (ring && P.Nq >= 1024));
}
template <int DT>
static void a_launch_dq(const AParams& P, const half_t* o, int ldo, const half_t* dO, int lddo, const float* lse,
                        float* Dbuf, half_t* dq, int lddq, hipStream_t s) {
    if constexpr (DT == 3 || DT == 5) {
        if (attn_bwd_ring_wanted(P) && lddo % 8 == 0 && ldo % 8 == 0) {
            constexpr int QT = DT == 3 ? 4 : 2;
            size_t smem = (size_t)3 * 2 * RingImage<DT>::BYTES;
            allow_big_smem(attn_bwd_dq_ring_kernel<DT, QT>, smem);
            dim3 grid = attn_grid(P, (P.Nq + 64 * QT - 1) / (64 * QT));
            MC_LAUNCH((attn_bwd_dq_ring_kernel<DT, QT>), grid, dim3(256), smem, s, P, o, ldo, dO, lddo, lse, Dbuf, dq, lddq);
            g_attn_last |= 1;
            return;
        }
    }
    if constexpr (DT == 3) {   // four row tiles per wave
        if (bwd_tiles(P.Nq, DT) == 4) return a_launch_dq_cfg<DT, 4>(P, o, ldo, dO, lddo, lse, Dbuf, dq, lddq, s);
    }
    if constexpr (DT <= 5) {
        if (bwd_tiles(P.Nq, DT) >= 2) return a_launch_dq_cfg<DT, 2>(P, o, ldo, dO, lddo, lse, Dbuf, dq, lddq, s);
    }
    a_launch_dq_cfg<DT, 1>(P, o, ldo, dO, lddo, lse, Dbuf, dq, lddq, s);
}
template <int DT>
static void a_launch_dkdv(const AParams& P, const half_t* dO, int lddo, const float* lse, const float* Dbuf,
                          half_t* dk, int lddk, half_t* dv, int lddv, hipStream_t s) {
    if constexpr (DT == 3 || DT == 5) {
        if (attn_bwd_ring_wanted(P) && lddo % 8 == 0) {
            constexpr int KT = DT == 3 ? MC_ATTN_DKDV_KT : 2;
            size_t smem = (size_t)3 * (2 * RingImage<DT>::BYTES + 2048);
            allow_big_smem(attn_bwd_dkdv_ring_kernel<DT, KT>, smem);
            dim3 grid = attn_grid(P, (P.Nk + 64 * KT - 1) / (64 * KT));
            MC_LAUNCH((attn_bwd_dkdv_ring_kernel<DT, KT>), grid, dim3(256), smem, s, P, dO, lddo, lse, Dbuf, dk, lddk, dv, lddv);
            g_attn_last |= 2;
            return;
        }
    }
    if constexpr (DT == 3) {
        if (bwd_tiles(P.Nk, DT) == 4) return a_launch_dkdv_cfg<DT, 4>(P, dO, lddo, lse, Dbuf, dk, lddk, dv, lddv, s);
    }
    if constexpr (DT <= 5) {
        if (bwd_tiles(P.Nk, DT) >= 2) return a_launch_dkdv_cfg<DT, 2>(P, dO, lddo, lse, Dbuf, dk, lddk, dv, lddv, s);
    }
    a_launch_dkdv_cfg<DT, 1>(P, dO, lddo, lse, Dbuf, dk, lddk, dv, lddv, s);
}

#define MC_A_DISPATCH(CALL)              \
    switch (dt) {                        \
        case 1: CALL(1); break;          \
        case 2: CALL(2); break;          \
        case 3: CALL(3); break;          \
        case 4: CALL(4); break;          \
        case 5: CALL(5); break;          \
        case 10: CALL(10); break;        \
        default: return MC_ERR_UNSUPPORTED; \
    }

}  // namespace mc

using namespace mc;

static AParams a_params(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, int Nq, int Nk,
                        int heads, int d, int nbatch, int kv_bdiv, float scale) {
    AParams P;
    P.q = (const half_t*)q; P.k = (const half_t*)k; P.v = (const half_t*)v;
    P.ldq = ldq; P.ldk = ldk; P.ldv = ldv; P.Nq = Nq; P.Nk = Nk; P.heads = heads; P.d = d;
    P.nbatch = nbatch; P.kv_bdiv = kv_bdiv; P.scale = scale; P.causal = 0;
    // one (batch, head) per XCD at a time pays when several row blocks share a K / V worth caching and there are units for all
    // 8 XCDs; MC_ATTN_XCD=0 restores the plain grid (A/B), 2 takes the mapping at every size (tests)
    const int xcd_env = MC_ENV_INT("MC_ATTN_XCD", 1);   // simulator / tools builds: re-read per call (the tests flip it)
    P.xcd_map = xcd_env == 2 || (xcd_env && heads * nbatch >= 8 && Nq >= 512 && Nk >= 512);
    return P;
}

// o[tokens, ldo], lse float[nbatch][heads][Nq] (may be null)
extern "C" int mc_attn_fwd_f16(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, void* o,
                               int ldo, float* lse, int Nq, int Nk, int heads, int d, int nbatch, int kv_bdiv,
                               float scale, void* stream) {
    AParams P = a_params(q, k, v, ldq, ldk, ldv, Nq, Nk, heads, d, nbatch, kv_bdiv, scale);
    if (!a_check(P) || ldo % 4) return MC_ERR_SHAPE;
    int dt = (d + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
    g_attn_last = 0;
#define CALL(DT_) a_launch_fwd<DT_>(P, (half_t*)o, ldo, lse, s)
    MC_A_DISPATCH(CALL)
#undef CALL
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

extern "C" int mc_attn_last_kernel(void) { return g_attn_last; }

// causal self-attention forward (CLIP text encoder): same contract, key j masked for query i when j > i
extern "C" int mc_attn_fwd_causal_f16(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, void* o,
                                      int ldo, int N, int heads, int d, int nbatch, float scale, void* stream) {
    AParams P = a_params(q, k, v, ldq, ldk, ldv, N, N, heads, d, nbatch, 1, scale);
    P.causal = 1;
    if (!a_check(P) || ldo % 4) return MC_ERR_SHAPE;
    int dt = (d + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
#define CALL(DT_) a_launch_fwd<DT_>(P, (half_t*)o, ldo, nullptr, s)
    MC_A_DISPATCH(CALL)
#undef CALL
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

// dq always; dk/dv only when non-null (self-attention).  Dbuf: float[nbatch][heads][Nq] workspace.
extern "C" int mc_attn_bwd_f16(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv,
                               const void* o, int ldo, const void* dO, int lddo, const float* lse, float* Dbuf,
                               void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, int Nq, int Nk,
                               int heads, int d, int nbatch, int kv_bdiv, float scale, void* stream) {
    AParams P = a_params(q, k, v, ldq, ldk, ldv, Nq, Nk, heads, d, nbatch, kv_bdiv, scale);
    if (!a_check(P) || ldo % 4 || lddo % 8 || lddq % 4 || !lse || !Dbuf) return MC_ERR_SHAPE;
    if ((dk || dv) && (!dk || !dv || kv_bdiv != 1 || lddk % 4 || lddv % 4)) return MC_ERR_SHAPE;
    int dt = (d + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
    g_attn_last = 0;
#define CALL(DT_)                                                                                             \
    a_launch_dq<DT_>(P, (const half_t*)o, ldo, (const half_t*)dO, lddo, lse, Dbuf, (half_t*)dq, lddq, s);     \
    if (dk) a_launch_dkdv<DT_>(P, (const half_t*)dO, lddo, lse, Dbuf, (half_t*)dk, lddk, (half_t*)dv, lddv, s)
    MC_A_DISPATCH(CALL)
#undef CALL
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}
