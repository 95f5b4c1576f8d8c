// Temporal self-attention of the AnimateDiff motion modules and the MotionClone
// guidance read-out on it.
// Reference: motion_module.py:274-345 (attention over the F frames of one spatial
// position), motionclone_functions.py:260-283 (P = softmax(scale q k^T)), :79
// (top-1 value/index), :85-100 (gather + MSE), :236 (gradient).
//
// Layout: activations stay in token order [(b f) (h w), C]; the reference's
// "(b f) d c -> (b d) f c" rearranges are index math here: the sequence of unit
// (b, p, head) is rows {(b*F + f)*HW + p} at column offset head*d.
// One wave per unit; the F x F score tile lives in MFMA accumulators
// (v_mfma_f32_16x16x16_f16, NT = ceil(F/16) tiles per side).  The transposed score
// tile S^T = K Q^T is computed so that softmax statistics are per-lane scalars
// (query = lane & 15) and P^T in accumulator layout is directly the B operand of
// O^T = V^T P^T - no cross-lane data movement for P.
// The backward kernel fuses: recompute P, the guidance-loss seed
// dP[q, idx[q]] += coef * (P[q, idx[q]] - ref[q]), softmax backward, and the three
// operand gradients; both tile orientations are produced by swapping MFMA operands.
#include "mc_common.hpp"
#include <cstdlib>

namespace mc {

// K = 16 step as the K = 32 instruction with the upper k-slots zero (the legacy v_mfma_f32_16x16x16_f16 returned wrong sums
// behind K = 32 steps on one accumulator in attention.hip; it is not used in this library).
// Round-2 finding (tools/race_dump.py, race_variants.py; profiles/r02_concurrency_and_determinism.md): with hipcc's SLP
// vectoriser on, the softmax-backward arithmetic of tattn_bwd_kernel became v_pk_mul / v_pk_fma / v_pk_add_f32 chains, and the
// row term D = sum(P * dP) of 30-60 of 32768 (pixel, head) units came out with one lane group's contribution missing whenever
// MFMA-heavy waves of ANOTHER stream (flash attention, 128x128 GEMM) shared the SIMD; inputs and every other intermediate were
// bit-identical, the exchange instruction (ds_bpermute / v_permlane*_swap) and SGPR operands made no difference, and the same
// source built with -fno-slp-vectorize (no packed fp32 VALU instructions) is bit-stable.  The library is built that way
// (build.py); it is also 1.4 % faster end to end.
__device__ __forceinline__ f32x4 mfma16z(half4_t a, half4_t b, f32x4 c) { return mfma16k32(cat4(a, zero4()), cat4(b, zero4()), c); }

#ifndef MC_TATTN_PROBE
#define MC_TATTN_PROBE 0
#endif

struct TParams {
    const half_t* q;
    const half_t* k;
    const half_t* v;
    int ld;       // row stride of q/k/v (elements)
    int B, F, HW, heads, d;
    float scale;
};

struct TUnit {
    int b, p, h;
    bool live;
};

__device__ __forceinline__ TUnit t_unit(const TParams& P) {
    int wave = threadIdx.x >> 6;
    long u = (long)blockIdx.x * (blockDim.x >> 6) + wave;
    TUnit r;
    long units = (long)P.B * P.HW * P.heads;
    r.live = u < units;
    if (!r.live) u = 0;
    r.h = (int)(u % P.heads);
    long bp = u / P.heads;
    r.p = (int)(bp % P.HW);
    r.b = (int)(bp / P.HW);
    return r;
}

__device__ __forceinline__ size_t t_row(const TParams& P, const TUnit& u, int f) {
    return ((size_t)u.b * P.F + f) * P.HW + u.p;
}

// row operand: X[f = 16*t + (lane&15)][16*ks + 4*(lane>>4) .. +4]
__device__ __forceinline__ half4_t t_row_frag(const half_t* x, int ld, const TParams& P, const TUnit& u, int t,
                                              int ks, int lane) {
    int f = 16 * t + (lane & 15);
    int c = 16 * ks + 4 * (lane >> 4);
    if (f < P.F && c < P.d) return ld4(x + t_row(P, u, f) * ld + u.h * P.d + c);
    return zero4();
}
// column operand: X^T[c = 16*dt + (lane&15)][f = 16*t + 4*(lane>>4) + j]
__device__ __forceinline__ half4_t t_col_frag(const half_t* x, int ld, const TParams& P, const TUnit& u, int t,
                                              int dt, int lane) {
    half4_t r;
    int c = 16 * dt + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int f = 16 * t + 4 * (lane >> 4) + j;
        r[j] = (f < P.F && c < P.d) ? x[t_row(P, u, f) * ld + u.h * P.d + c] : (half_t)0.f;
    }
    return r;
}

// exchanges across the 4 lane groups sharing lane&15: v_permlane16_swap / v_permlane32_swap (gfx950; VALU only, no trip
// through the LDS crossbar as with ds_bpermute).  swap(a, a) leaves {rows 0,0,2,2} / {rows 1,1,3,3} resp. {lo,lo} / {hi,hi}.
#ifndef MC_EMU
__device__ __forceinline__ float xchg16(float v) {   // v[lane ^ 16]
    unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 16) ? r[0] : r[1]);
}
__device__ __forceinline__ float xchg32(float v) {   // v[lane ^ 32]
    unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}
#else
__device__ inline float xchg16(float v) { return shfl_xor(v, 16); }
__device__ inline float xchg32(float v) { return shfl_xor(v, 32); }
#endif
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, xchg16(v));
    return fmaxf(v, xchg32(v));
}
__device__ __forceinline__ float group_sum(float v) {
    v += xchg16(v);
    return v + xchg32(v);
}

// Scores of query tile tq against all key tiles, transposed orientation:
// st[tk][i] = scale * S[q = 16tq + (lane&15)][kv = 16tk + 4g + i]; invalid kv -> -inf
template <int NT, int DT>
__device__ __forceinline__ void t_scores_T(const TParams& P, const TUnit& u, int tq, int lane, f32x4 (&st)[NT]) {
#pragma unroll
    for (int tk = 0; tk < NT; ++tk) st[tk] = fzero4();
#pragma unroll
    for (int ks = 0; ks < DT; ++ks) {
        half4_t qf = t_row_frag(P.q, P.ld, P, u, tq, ks, lane);
#pragma unroll
        for (int tk = 0; tk < NT; ++tk) {
            half4_t kf = t_row_frag(P.k, P.ld, P, u, tk, ks, lane);
            st[tk] = mfma16z(kf, qf, st[tk]);
        }
    }
#pragma unroll
    for (int tk = 0; tk < NT; ++tk)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int kv = 16 * tk + 4 * (lane >> 4) + i;
            st[tk][i] = kv < P.F ? st[tk][i] * P.scale : -INFINITY;
        }
}

// softmax over kv for the lane's query column; returns (max, sum) and leaves exp(s - m) in st
template <int NT>
__device__ __forceinline__ void t_softmax_T(f32x4 (&st)[NT], float& m, float& l) {
    m = -INFINITY;
#pragma unroll
    for (int tk = 0; tk < NT; ++tk)
#pragma unroll
        for (int i = 0; i < 4; ++i) m = fmaxf(m, st[tk][i]);
    m = group_max(m);
    l = 0.f;
#pragma unroll
    for (int tk = 0; tk < NT; ++tk)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float e = expf(st[tk][i] - m);
            st[tk][i] = e;
            l += e;
        }
    l = group_sum(l);
}

// ---- forward --------------------------------------------------------------------------------
// mode 0: write attention output o; mode 1: write top-1 (value fp16, index u8) of P per query;
// mode 2: write the per-query squared error (P[q, idx[q]] - ref[q])^2 summed per unit.
template <int NT, int DT>
__global__ __launch_bounds__(256) void tattn_fwd_kernel(TParams P, half_t* o, int ldo, int mode,
                                                         half_t* top_val, uint8_t* top_idx,
                                                         const uint8_t* ref_idx, const float* ref_val,
                                                         float* unit_loss) {
    const int lane = threadIdx.x & 63;
    TUnit u = t_unit(P);
    if (!u.live) return;  // whole wave
    const long unit = ((long)u.b * P.HW + u.p) * P.heads + u.h;
    float loss_acc = 0.f;
#pragma unroll
    for (int tq = 0; tq < NT; ++tq) {
        f32x4 st[NT];
        t_scores_T<NT, DT>(P, u, tq, lane, st);
        const int qf = 16 * tq + (lane & 15);
        if (mode == 1 || mode == 3) {
            // The reference's order of operations in its own (fp16) arithmetic (attention.py:593-609,
            // motionclone_functions.py:79): scores leave baddbmm ROUNDED TO fp16 (fp32 accumulate, alpha = scale applied
            // before the rounding); softmax is evaluated in fp32 on those fp16 scores - exp(s - max) / sum, a true
            // division - and ROUNDED TO fp16; topk(k = 1) then runs on the fp16 probabilities, so distinct scores whose
            // probabilities round to the same fp16 value are ties.  Ties go to the lowest index.
#pragma unroll
            for (int tk = 0; tk < NT; ++tk)
#pragma unroll
                for (int i = 0; i < 4; ++i) st[tk][i] = (float)(half_t)st[tk][i];   // -inf (masked kv) stays -inf
            float m = -INFINITY;
#pragma unroll
            for (int tk = 0; tk < NT; ++tk)
#pragma unroll
                for (int i = 0; i < 4; ++i) m = fmaxf(m, st[tk][i]);
            m = group_max(m);
            // exp / sum / quotient in double: the library is built with fast-math, whose fp32 exp (error ~ |x| 2^-24) and
            // reciprocal would move values across fp16 rounding boundaries that the reference's accurate fp32 softmax
            // does not cross; this read-out runs once per video on 6 x [256, 8, 16, 16] scores
            double ed[NT][4], l = 0.0;
#pragma unroll
            for (int tk = 0; tk < NT; ++tk)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ed[tk][i] = st[tk][i] == -INFINITY ? 0.0 : exp((double)st[tk][i] - (double)m);
                    l += ed[tk][i];
                }
            {   // sum over the 4 lane groups that share this query
                float lo = (float)l, hi = (float)(l - (double)lo);
                lo = group_sum(lo);
                hi = group_sum(hi);
                l = (double)lo + (double)hi;
            }
            float best = -1.f;
            int bi = 0;
#pragma unroll
            for (int tk = 0; tk < NT; ++tk)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int kv = 16 * tk + 4 * (lane >> 4) + i;
                    half_t ph = (half_t)(float)(ed[tk][i] / l);
                    if (mode == 3) {
                        if (qf < P.F && kv < P.F) top_val[(unit * P.F + qf) * P.F + kv] = ph;
                    } else if (kv < P.F && (float)ph > best) {   // ascending kv within the lane: first maximum kept
                        best = (float)ph;
                        bi = kv;
                    }
                }
            if (mode == 3) continue;
#pragma unroll
            for (int msk = 16; msk <= 32; msk <<= 1) {
                float ob = shfl_xor(best, msk);
                int oi = shfl_xor(bi, msk);
                if (ob > best || (ob == best && oi < bi)) {
                    best = ob;
                    bi = oi;
                }
            }
            if (lane < 16 && qf < P.F) {
                top_val[unit * P.F + qf] = (half_t)best;
                top_idx[unit * P.F + qf] = (uint8_t)bi;
            }
            continue;
        }
        float m, l;
        t_softmax_T<NT>(st, m, l);
        const float inv = 1.0f / l;
        if (mode == 2) {
            int idx = qf < P.F ? (int)ref_idx[unit * P.F + qf] : 0;
            float pv = 0.f;
#pragma unroll
            for (int tk = 0; tk < NT; ++tk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (16 * tk + 4 * (lane >> 4) + i == idx) pv = st[tk][i] * inv;
            pv = group_sum(pv);
            if (lane < 16 && qf < P.F) {
                float dlt = pv - ref_val[unit * P.F + qf];
                loss_acc += dlt * dlt;
            }
            continue;
        }
        // O^T[dt] = sum_tk V^T[dt][tk] * P^T[tk]
        half4_t pf[NT];
#pragma unroll
        for (int tk = 0; tk < NT; ++tk)
#pragma unroll
            for (int i = 0; i < 4; ++i) pf[tk][i] = (half_t)(st[tk][i] * inv);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            f32x4 acc = fzero4();
#pragma unroll
            for (int tk = 0; tk < NT; ++tk) {
                half4_t vf = t_col_frag(P.v, P.ld, P, u, tk, dt, lane);
                acc = mfma16z(vf, pf[tk], acc);
            }
            int c = 16 * dt + 4 * (lane >> 4);
            if (qf < P.F && c < P.d) {
                half4_t ov;
#pragma unroll
                for (int i = 0; i < 4; ++i) ov[i] = to_half(acc[i]);
                st4(o + t_row(P, u, qf) * ldo + u.h * P.d + c, ov);
            }
        }
    }
    if (mode == 2) {
        loss_acc = wave_sum(loss_acc);
        if (lane == 0) unit_loss[unit] = loss_acc;
    }
}

// ---- forward, attention output only, 16-byte loads (d % 8 == 0) --------------------------------------------------------------
// The kernel above reads its operands 8 bytes per lane in DT steps and gathers V^T two bytes at a time (4 DT scalar loads per
// lane): 18 load instructions for 3.75 KB per unit at d = 40, and it ran at 2.4 TB/s.  Here every operand row is read once,
// 16 bytes per lane: the contraction index of the K = 32 MFMA is permuted so that lane group g's 8 k-slots of step s are the 8
// adjacent head-dim elements [32 s + 8 g, +8) - 2 instructions per operand at d = 40, and 2 MFMAs per score tile instead of 3.
// V goes through a wave-private row-major LDS image and comes back as V^T fragments by the hardware transpose read.
template <int NT, int NS>
__global__ __launch_bounds__(256) void tattn_fwd_vec_kernel(TParams P, half_t* o, int ldo) {
    constexpr int PB = 64 * NS + 16;          // LDS row pitch, bytes (head dim padded to 32 NS)
    constexpr int WAVE_LDS = 16 * NT * PB;
    __shared__ __attribute__((aligned(16))) char vimg[4 * WAVE_LDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c15 = lane & 15;
    TUnit u = t_unit(P);
    if (!u.live) return;  // whole wave
    char* vl = vimg + wave * WAVE_LDS;
    const int nch = P.d / 8;                  // 16-byte chunks per row
    half8_t qf[NT][NS], kf[NT][NS];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int f = 16 * t + c15;
        const size_t row = t_row(P, u, f < P.F ? f : 0) * P.ld + u.h * P.d;
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            const int ch = 4 * s2 + g;
            const bool ok = f < P.F && ch < nch;
            qf[t][s2] = ok ? ld8(P.q + row + 8 * ch) : zero8();
            kf[t][s2] = ok ? ld8(P.k + row + 8 * ch) : zero8();
            *reinterpret_cast<half8_t*>(vl + f * PB + 16 * ch) = ok ? ld8(P.v + row + 8 * ch) : zero8();
        }
    }
    wave_lds_sync();
    constexpr int DT = 2 * NS;                // 16-wide output tiles (those at / beyond d are skipped)
#pragma unroll
    for (int tq = 0; tq < NT; ++tq) {
        f32x4 st[NT];
#pragma unroll
        for (int tk = 0; tk < NT; ++tk) {
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) st[tk] = mfma16k32(kf[tk][s2], qf[tq][s2], s2 == 0 ? fzero4() : st[tk]);
#pragma unroll
            for (int i = 0; i < 4; ++i) st[tk][i] = 16 * tk + 4 * g + i < P.F ? st[tk][i] * P.scale : -INFINITY;
        }
        float m, l;
        t_softmax_T<NT>(st, m, l);
        const float inv = 1.0f / l;
        // P^T as the B operand; with two key tiles both go into ONE K = 32 step (k-slots 0-3: tile 0, 4-7: tile 1)
        half8_t pf;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pf[i] = (half_t)(st[0][i] * inv);
            pf[4 + i] = NT == 2 ? (half_t)(st[NT - 1][i] * inv) : (half_t)0.f;
        }
        const int qrow = 16 * tq + c15;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            if (16 * dt >= P.d) break;   // uniform
            // V^T[c = 16 dt + c15][f = 4 g + j (+ 16)]: lane passes &V[4 g + (c15 >> 2)][16 dt + 4 (c15 & 3)]
            const half_t* vp = reinterpret_cast<const half_t*>(vl + (4 * g + (c15 >> 2)) * PB + 32 * dt + 8 * (c15 & 3));
            const half8_t vf = cat4(lds_read_tr4(vp), NT == 2 ? lds_read_tr4(vp + 8 * PB) : zero4());
            const f32x4 acc = mfma16k32(vf, pf, fzero4());
            const int c = 16 * dt + 4 * g;
            if (qrow < P.F && c < P.d) {
                half4_t ov;
#pragma unroll
                for (int i = 0; i < 4; ++i) ov[i] = to_half(acc[i]);
                st4(o + t_row(P, u, qrow) * ldo + u.h * P.d + c, ov);
            }
        }
    }
}

// ---- backward -------------------------------------------------------------------------------
// inputs: q,k,v (P), dO (may be null), guidance seed (ref_idx/ref_val may be null, coef)
// outputs: dq, dk, dv with row stride ldg (same token layout)
// VEC (round 3; d % 8 == 0, 16-byte aligned rows): every operand row is read ONCE, 16 bytes per lane (k-slot permutation of
// tattn_fwd_vec_kernel: the K = 32 steps add the same products in the same order as the zero-extended K = 16 steps, bit for bit),
// and Q, K, dO are kept as wave-private row-major LDS images from which the gradient MFMAs take their transposed fragments by
// ds_read_b64_tr_b16 - 8 load instructions per unit at d = 40 instead of 48 (36 of them two-byte gathers).
template <int NT, int DT, int VAR = 0, bool VEC = false>
__global__ __launch_bounds__(256) void tattn_bwd_kernel(TParams P, const half_t* dO, int lddo, half_t* dq,
                                                         half_t* dk, half_t* dv, int ldg,
                                                         const uint8_t* ref_idx, const float* ref_val,
                                                         float seed_coef, float* dbg = nullptr) {
    constexpr int NS = (DT + 1) / 2;
    constexpr int PB = 64 * NS + 16;            // VEC: image row pitch, bytes
    constexpr int IMG = 16 * NT * PB;           // one image of one wave
    MC_DYN_SMEM(smem);
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, c15 = lane & 15;
    TUnit u = t_unit(P);
    if (!u.live) return;
    const long unit = ((long)u.b * P.HW + u.p) * P.heads + u.h;
    char* img_q = smem + (threadIdx.x >> 6) * 3 * IMG;   // VEC: Q, K, dO rows of this unit
    char* img_k = img_q + IMG;
    char* img_o = img_k + IMG;
    const int nch = P.d / 8;

    // S (q rows), S^T (kv rows), dP, dP^T accumulated over the head dimension
    f32x4 s[NT][NT], sT[NT][NT], dp[NT][NT], dpT[NT][NT];  // [tq][tk]
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) s[a][b] = sT[a][b] = dp[a][b] = dpT[a][b] = fzero4();
    // two passes over the head dimension, two accumulator chains each (scores, then dP)
    if constexpr (VEC) {
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            half8_t qf[NT], kf[NT];
            const int ch = 4 * s2 + g;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int f = 16 * t + c15;
                const bool ok = f < P.F && ch < nch;
                const size_t row = t_row(P, u, f < P.F ? f : 0) * P.ld + u.h * P.d + 8 * ch;
                qf[t] = ok ? ld8(P.q + row) : zero8();
                kf[t] = ok ? ld8(P.k + row) : zero8();
                *reinterpret_cast<half8_t*>(img_q + f * PB + 16 * ch) = qf[t];
                *reinterpret_cast<half8_t*>(img_k + f * PB + 16 * ch) = kf[t];
            }
#pragma unroll
            for (int tq = 0; tq < NT; ++tq)
#pragma unroll
                for (int tk = 0; tk < NT; ++tk) {
                    s[tq][tk] = mfma16k32(qf[tq], kf[tk], s[tq][tk]);
                    sT[tq][tk] = mfma16k32(kf[tk], qf[tq], sT[tq][tk]);
                }
        }
        if (dO) {
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                half8_t vf[NT], of[NT];
                const int ch = 4 * s2 + g;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int f = 16 * t + c15;
                    const bool ok = f < P.F && ch < nch;
                    const size_t base = t_row(P, u, f < P.F ? f : 0);
                    vf[t] = ok ? ld8(P.v + base * P.ld + u.h * P.d + 8 * ch) : zero8();
                    of[t] = ok ? ld8(dO + base * lddo + u.h * P.d + 8 * ch) : zero8();
                    *reinterpret_cast<half8_t*>(img_o + f * PB + 16 * ch) = of[t];
                }
#pragma unroll
                for (int tq = 0; tq < NT; ++tq)
#pragma unroll
                    for (int tk = 0; tk < NT; ++tk) {
                        dp[tq][tk] = mfma16k32(of[tq], vf[tk], dp[tq][tk]);
                        dpT[tq][tk] = mfma16k32(vf[tk], of[tq], dpT[tq][tk]);
                    }
            }
        }
        wave_lds_sync();
    } else {
#pragma unroll
        for (int ks = 0; ks < DT; ++ks) {
            half4_t qf[NT], kf[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                qf[t] = t_row_frag(P.q, P.ld, P, u, t, ks, lane);
                kf[t] = t_row_frag(P.k, P.ld, P, u, t, ks, lane);
            }
#pragma unroll
            for (int tq = 0; tq < NT; ++tq)
#pragma unroll
                for (int tk = 0; tk < NT; ++tk) {
                    s[tq][tk] = mfma16z(qf[tq], kf[tk], s[tq][tk]);    // [q = 4g+i][kv = c15]
                    sT[tq][tk] = mfma16z(kf[tk], qf[tq], sT[tq][tk]);  // [kv = 4g+i][q = c15]
                }
        }
        if (dO) {
#pragma unroll
            for (int ks = 0; ks < DT; ++ks) {
                half4_t vf[NT], of[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    vf[t] = t_row_frag(P.v, P.ld, P, u, t, ks, lane);
                    of[t] = t_row_frag(dO, lddo, P, u, t, ks, lane);
                }
#pragma unroll
                for (int tq = 0; tq < NT; ++tq)
#pragma unroll
                    for (int tk = 0; tk < NT; ++tk) {
                        dp[tq][tk] = mfma16z(of[tq], vf[tk], dp[tq][tk]);
                        dpT[tq][tk] = mfma16z(vf[tk], of[tq], dpT[tq][tk]);
                    }
            }
        }
    }

    // per-query statistics in the transposed orientation (query = 16tq + c15)
    float mq[NT], lq[NT], Dq[NT];
    int idxq[NT];
    float refq[NT];
#pragma unroll
    for (int tq = 0; tq < NT; ++tq) {
        const int qv = 16 * tq + c15;
        idxq[tq] = (ref_idx && qv < P.F) ? (int)ref_idx[unit * P.F + qv] : -1;
        refq[tq] = (ref_idx && qv < P.F) ? ref_val[unit * P.F + qv] : 0.f;
        float m = -INFINITY;
#pragma unroll
        for (int tk = 0; tk < NT; ++tk)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int kv = 16 * tk + 4 * g + i;
                sT[tq][tk][i] = kv < P.F ? sT[tq][tk][i] * P.scale : -INFINITY;
                m = fmaxf(m, sT[tq][tk][i]);
            }
        m = group_max(m);
        float l = 0.f;
#pragma unroll
        for (int tk = 0; tk < NT; ++tk)
#pragma unroll
            for (int i = 0; i < 4; ++i) l += expf(sT[tq][tk][i] - m);
        l = group_sum(l);
        mq[tq] = m;
        lq[tq] = l;
        // P^T, total dP^T (attention path + guidance seed), D = sum_kv P * dP
        float dsum = 0.f;
#pragma unroll
        for (int tk = 0; tk < NT; ++tk)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int kv = 16 * tk + 4 * g + i;
                float pv = expf(sT[tq][tk][i] - m) / l;
                float d = dpT[tq][tk][i];
                if (kv == idxq[tq]) d += seed_coef * (pv - refq[tq]);
#if MC_TATTN_PROBE & 1   // tools/tattn_race.py: which packed-fp32 chain of an SLP build goes wrong (see header)
                pv = opaque(pv);
#endif
#if MC_TATTN_PROBE & 2
                d = opaque(d);
#endif
                sT[tq][tk][i] = pv;
                dpT[tq][tk][i] = d;
                dsum += pv * d;
#if MC_TATTN_PROBE & 4
                dsum = opaque(dsum);
#endif
            }
        Dq[tq] = group_sum(dsum);
    }

    // dS^T = P^T * (dP^T - D) (fp16 B operands); and the q-row orientation via lane broadcasts
    half4_t dsT[NT][NT], ds[NT][NT], pr[NT][NT];
#pragma unroll
    for (int tq = 0; tq < NT; ++tq) {
#pragma unroll
        for (int tk = 0; tk < NT; ++tk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                dsT[tq][tk][i] = (half_t)(sT[tq][tk][i] * (dpT[tq][tk][i] - Dq[tq]));
        // rows q = 16tq + 4g + i: fetch (m, l, D, idx, ref) from the lane whose column is that query
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int src = 4 * g + i;
            float m = shfl(mq[tq], src), l = shfl(lq[tq], src), D = shfl(Dq[tq], src);
            int idx = shfl(idxq[tq], src);
            float rf = shfl(refq[tq], src);
            const bool qok = 16 * tq + src < P.F;
#pragma unroll
            for (int tk = 0; tk < NT; ++tk) {
                int kv = 16 * tk + c15;
                float pv = 0.f, d = 0.f;
                if (qok && kv < P.F) {
                    pv = expf(s[tq][tk][i] * P.scale - m) / l;
                    d = dp[tq][tk][i];
                    if (kv == idx) d += seed_coef * (pv - rf);
                }
                pr[tq][tk][i] = (half_t)pv;
                ds[tq][tk][i] = (half_t)(pv * (d - D));
            }
        }
    }

    if constexpr (VAR == 1) {   // tools/race_dump.py: intermediates of tile (0, 0), 24 floats per lane
        float* o = dbg + ((size_t)unit * 64 + lane) * 24;
        o[0] = mq[0]; o[1] = lq[0]; o[2] = Dq[0]; o[3] = (float)idxq[0]; o[4] = refq[0];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[5 + i] = sT[0][0][i];
            o[9 + i] = dpT[0][0][i];
            o[13 + i] = (float)dsT[0][0][i];
            o[17 + i] = (float)ds[0][0][i];
        }
        o[21] = (float)pr[0][0][0]; o[22] = s[0][0][0]; o[23] = dp[0][0][0];
    }

    // operand gradients, one 16-wide slice of the head dimension at a time
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const int c = 16 * dt + 4 * g;
        half4_t kc[NT], qc[NT], oc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if constexpr (VEC) {
                // X^T[c = 16 dt + c15][f = 16 t + 4 g + j]: the lane passes &X[16 t + 4 g + (c15 >> 2)][16 dt + 4 (c15 & 3)]
                if (16 * dt < P.d) {   // uniform (tiles at / beyond d have no image columns when d % 32 == 0)
                    const int off = (16 * t + 4 * g + (c15 >> 2)) * PB + 32 * dt + 8 * (c15 & 3);
                    kc[t] = lds_read_tr4(reinterpret_cast<const half_t*>(img_k + off));
                    qc[t] = lds_read_tr4(reinterpret_cast<const half_t*>(img_q + off));
                    if (dO) oc[t] = lds_read_tr4(reinterpret_cast<const half_t*>(img_o + off));
                } else {
                    kc[t] = qc[t] = oc[t] = zero4();
                }
            } else {
                kc[t] = t_col_frag(P.k, P.ld, P, u, t, dt, lane);
                qc[t] = t_col_frag(P.q, P.ld, P, u, t, dt, lane);
                if (dO) oc[t] = t_col_frag(dO, lddo, P, u, t, dt, lane);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // dQ^T[d][q] = sum_kv K^T[d][kv] dS^T[kv][q]
            f32x4 aq = fzero4(), ak = fzero4(), av = fzero4();
#pragma unroll
            for (int t2 = 0; t2 < NT; ++t2) {
                aq = mfma16z(kc[t2], dsT[t][t2], aq);
                ak = mfma16z(qc[t2], ds[t2][t], ak);       // dK^T[d][kv] = sum_q Q^T[d][q] dS[q][kv]
                if (dO) av = mfma16z(oc[t2], pr[t2][t], av);  // dV^T[d][kv] = sum_q dO^T[d][q] P[q][kv]
            }
            const int f = 16 * t + c15;
            if (f < P.F && c < P.d) {
                half4_t oq, ok, ov;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    oq[i] = to_half(aq[i] * P.scale);
                    ok[i] = to_half(ak[i] * P.scale);
                    ov[i] = to_half(av[i]);
                }
                size_t off = t_row(P, u, f) * ldg + u.h * P.d + c;
                st4(dq + off, oq);
                st4(dk + off, ok);
                st4(dv + off, ov);
            }
        }
    }
}

__global__ void reduce_sum_kernel(const float* in, long n, float scale, float* out) {
    __shared__ float red[4];
    float acc = 0.f;
#pragma clang loop vectorize(disable)   // no packed-fp32 adds (tests/test_determinism.py), one fixed summation order
    for (long i = threadIdx.x; i < n; i += blockDim.x) acc += in[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) * scale;
}

static thread_local int g_tattn_last = 0;   // 1: the calling thread's last temporal-attention entry ran a 16-byte-load kernel
#ifdef MC_TOOLS
static float* g_tattn_debug_buf = nullptr;   // mc_tattn_debug_buffer: intermediates of the F <= 16, d = 40 backward (tools build only)
#endif

template <int NT, int DT>
static void t_launch_fwd(const TParams& P, half_t* o, int ldo, int mode, half_t* tv, uint8_t* ti,
                         const uint8_t* ri, const float* rv, float* ul, hipStream_t s) {
    long units = (long)P.B * P.HW * P.heads;
    g_tattn_last = 0;
    if constexpr (DT == 3 || DT == 5 || DT == 10) {
        // attention output, rows readable 16 bytes at a time (MC_TATTN_VEC=0: the 8-byte kernel, A/B)
        static const int vec_env = MC_ENV_INT("MC_TATTN_VEC", 1);
        const bool aligned = ((uintptr_t)P.q | (uintptr_t)P.k | (uintptr_t)P.v) % 16 == 0;
        if (mode == 0 && vec_env && P.d % 8 == 0 && P.ld % 8 == 0 && aligned) {
            MC_LAUNCH((tattn_fwd_vec_kernel<NT, (DT + 1) / 2>), dim3((unsigned)((units + 3) / 4)), dim3(256), 0, s, P, o, ldo);
            g_tattn_last = 1;
            return;
        }
    }
    MC_LAUNCH((tattn_fwd_kernel<NT, DT>), dim3((unsigned)((units + 3) / 4)), dim3(256), 0, s, P, o, ldo, mode, tv,
              ti, ri, rv, ul);
}
template <int NT, int DT>
static void t_launch_bwd(const TParams& P, const half_t* dO, int lddo, half_t* dq, half_t* dk, half_t* dv,
                         int ldg, const uint8_t* ri, const float* rv, float coef, hipStream_t s) {
    long units = (long)P.B * P.HW * P.heads;
    g_tattn_last = 0;
#ifdef MC_TOOLS
    if constexpr (NT == 1 && DT == 3) {
        if (g_tattn_debug_buf) {
            MC_LAUNCH((tattn_bwd_kernel<NT, DT, 1>), dim3((unsigned)((units + 3) / 4)), dim3(256), 0, s, P, dO, lddo,
                      dq, dk, dv, ldg, ri, rv, coef, g_tattn_debug_buf);
            return;
        }
    }
#endif
    if constexpr (DT == 3 || DT == 5 || DT == 10) {
        static const int vec_env = MC_ENV_INT("MC_TATTN_VEC", 1);
        const bool aligned = ((uintptr_t)P.q | (uintptr_t)P.k | (uintptr_t)P.v | (uintptr_t)dO) % 16 == 0;
        constexpr int NS = (DT + 1) / 2;
        constexpr size_t smem = (size_t)4 * 3 * 16 * NT * (64 * NS + 16);
        // (images of at most 48 KiB per workgroup: F <= 16 at d = 40 / 80; beyond that the occupancy lost costs more)
        if (smem <= 48 * 1024 && vec_env && P.d % 8 == 0 && P.ld % 8 == 0 && (!dO || lddo % 8 == 0) && aligned) {
            allow_big_smem(tattn_bwd_kernel<NT, DT, 0, true>, smem);
            MC_LAUNCH((tattn_bwd_kernel<NT, DT, 0, true>), dim3((unsigned)((units + 3) / 4)), dim3(256), smem, s, P, dO, lddo,
                      dq, dk, dv, ldg, ri, rv, coef, (float*)nullptr);
            g_tattn_last = 1;
            return;
        }
    }
    MC_LAUNCH((tattn_bwd_kernel<NT, DT>), dim3((unsigned)((units + 3) / 4)), dim3(256), 0, s, P, dO, lddo, dq, dk,
              dv, ldg, ri, rv, coef, (float*)nullptr);
}

#define MC_T_DISPATCH(CALL)                                                     \
    switch (nt * 16 + dt) {                                                     \
        case 1 * 16 + 1: CALL(1, 1); break;                                     \
        case 1 * 16 + 2: CALL(1, 2); break;                                     \
        case 1 * 16 + 3: CALL(1, 3); break;                                     \
        case 1 * 16 + 4: CALL(1, 4); break;                                     \
        case 1 * 16 + 5: CALL(1, 5); break;                                     \
        case 1 * 16 + 10: CALL(1, 10); break;                                   \
        case 2 * 16 + 1: CALL(2, 1); break;                                     \
        case 2 * 16 + 2: CALL(2, 2); break;                                     \
        case 2 * 16 + 3: CALL(2, 3); break;                                     \
        case 2 * 16 + 4: CALL(2, 4); break;                                     \
        case 2 * 16 + 5: CALL(2, 5); break;                                     \
        case 2 * 16 + 10: CALL(2, 10); break;                                   \
        default: return MC_ERR_UNSUPPORTED;                                     \
    }

static int t_check(const TParams& P) {
    if (P.B <= 0 || P.F <= 0 || P.HW <= 0 || P.heads <= 0 || P.d <= 0) return 0;
    if (P.d % 4 || P.ld % 4 || P.F > 32) return 0;
    return 1;
}

}  // namespace mc

using namespace mc;

static TParams t_params(const void* q, const void* k, const void* v, int ld, int B, int F, int HW, int heads,
                        int d, float scale) {
    TParams P;
    P.q = (const half_t*)q; P.k = (const half_t*)k; P.v = (const half_t*)v;
    P.ld = ld; P.B = B; P.F = F; P.HW = HW; P.heads = heads; P.d = d; P.scale = scale;
    return P;
}

extern "C" int mc_tattn_last_kernel(void) { return g_tattn_last; }

extern "C" int mc_tattn_fwd_f16(const void* q, const void* k, const void* v, int ld, void* o, int ldo, int B,
                                int F, int HW, int heads, int d, float scale, void* stream) {
    TParams P = t_params(q, k, v, ld, B, F, HW, heads, d, scale);
    if (!t_check(P) || ldo % 4) return MC_ERR_SHAPE;
    int nt = (F + 15) / 16, dt = (d + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
#define CALL(NT_, DT_) t_launch_fwd<NT_, DT_>(P, (half_t*)o, ldo, 0, nullptr, nullptr, nullptr, nullptr, nullptr, s)
    MC_T_DISPATCH(CALL)
#undef CALL
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

// motion representation: top_val fp16 [B*HW, heads, F], top_idx u8 [B*HW, heads, F]
extern "C" int mc_tattn_top1_f16(const void* q, const void* k, int ld, void* top_val, void* top_idx, int B,
                                 int F, int HW, int heads, int d, float scale, void* stream) {
    TParams P = t_params(q, k, k, ld, B, F, HW, heads, d, scale);
    if (!t_check(P)) return MC_ERR_SHAPE;
    int nt = (F + 15) / 16, dt = (d + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
#define CALL(NT_, DT_) \
    t_launch_fwd<NT_, DT_>(P, nullptr, 0, 1, (half_t*)top_val, (uint8_t*)top_idx, nullptr, nullptr, nullptr, s)
    MC_T_DISPATCH(CALL)
#undef CALL
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

// motionclone_functions.py:260-283: prob fp16 [B*HW, heads, F, F]
extern "C" int mc_tattn_prob_f16(const void* q, const void* k, int ld, void* prob, int B, int F, int HW, int heads,
                                 int d, float scale, void* stream) {
    TParams P = t_params(q, k, k, ld, B, F, HW, heads, d, scale);
    if (!t_check(P)) return MC_ERR_SHAPE;
    int nt = (F + 15) / 16, dt = (d + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
#define CALL(NT_, DT_) \
    t_launch_fwd<NT_, DT_>(P, nullptr, 0, 3, (half_t*)prob, nullptr, nullptr, nullptr, nullptr, s)
    MC_T_DISPATCH(CALL)
#undef CALL
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

// loss[0] = mean over (unit, frame) of (P[q, idx] - ref)^2; workspace unit_loss float[B*HW*heads]
extern "C" int mc_tattn_loss_f16(const void* q, const void* k, int ld, const void* ref_idx,
                                 const float* ref_val, float* unit_loss, float* loss, int B, int F, int HW,
                                 int heads, int d, float scale, void* stream) {
    TParams P = t_params(q, k, k, ld, B, F, HW, heads, d, scale);
    if (!t_check(P)) return MC_ERR_SHAPE;
    int nt = (F + 15) / 16, dt = (d + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
#define CALL(NT_, DT_) \
    t_launch_fwd<NT_, DT_>(P, nullptr, 0, 2, nullptr, nullptr, (const uint8_t*)ref_idx, ref_val, unit_loss, s)
    MC_T_DISPATCH(CALL)
#undef CALL
    long units = (long)B * HW * heads;
    MC_LAUNCH(reduce_sum_kernel, dim3(1), dim3(256), 0, s, (const float*)unit_loss, units,
              1.0f / (float)(units * F), loss);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

extern "C" int mc_tattn_bwd_f16(const void* q, const void* k, const void* v, int ld, const void* dO, int lddo,
                                void* dq, void* dk, void* dv, int ldg, const void* ref_idx,
                                const float* ref_val, float seed_coef, int B, int F, int HW, int heads, int d,
                                float scale, void* stream) {
    TParams P = t_params(q, k, v, ld, B, F, HW, heads, d, scale);
    if (!t_check(P) || ldg % 4 || (dO && lddo % 4)) return MC_ERR_SHAPE;
    if (!dO && !ref_idx) return MC_ERR_SHAPE;
    int nt = (F + 15) / 16, dt = (d + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
#define CALL(NT_, DT_)                                                                                        \
    t_launch_bwd<NT_, DT_>(P, (const half_t*)dO, lddo, (half_t*)dq, (half_t*)dk, (half_t*)dv, ldg,          \
                           (const uint8_t*)ref_idx, ref_val, seed_coef, s)
    MC_T_DISPATCH(CALL)
#undef CALL
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

#ifdef MC_TOOLS
extern "C" int mc_tattn_debug_buffer(void* buf) {   // units * 64 * 24 floats, or null to switch the dump off
    g_tattn_debug_buf = (float*)buf;
    return 0;
}
#endif

extern "C" int mc_reduce_sum_f32(const float* in, long n, float scale, float* out, void* stream) {
    if (n <= 0) return MC_ERR_SHAPE;
    MC_LAUNCH(reduce_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, in, n, scale, out);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}
