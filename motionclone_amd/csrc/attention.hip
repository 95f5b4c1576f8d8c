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

namespace mc {

struct AParams {
    const half_t* q;
    const half_t* k;
    const half_t* v;
    int ldq, ldk, ldv;
    int Nq, Nk, heads, d, nbatch, kv_bdiv;
    float scale;
};

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
// same rows, stored transposed: dst[c][r] (row length TPAD)
__device__ __forceinline__ void stage_cols(half_t* dst, const half_t* src, size_t row_base, int r0, int nrows,
                                           int ld, int col0, int d) {
    const int vpr = d / 8;
    for (int idx = threadIdx.x; idx < KV_TILE * vpr; idx += blockDim.x) {
        int r = idx / vpr, vcol = idx - r * vpr;
        half8_t val = zero8();
        if (r0 + r < nrows) val = ld8(src + (row_base + r0 + r) * ld + col0 + vcol * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[(vcol * 8 + e) * TPAD + r] = val[e];
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
template <int DT, int QT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AParams P, half_t* o, int ldo, float* lse) {
    constexpr int RP = DT * 16 + 8;
    MC_DYN_SMEM(smem);
    half_t* Ks = reinterpret_cast<half_t*>(smem);  // [64][RP]
    half_t* Vt = Ks + KV_TILE * RP;                // [DT*16][TPAD]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c15 = lane & 15;
    const int h = blockIdx.y, b = blockIdx.z;
    const int col0 = h * P.d;
    const size_t qbase = (size_t)b * P.Nq, kbase = (size_t)(b / P.kv_bdiv) * P.Nk;
    const int q0 = blockIdx.x * (64 * QT) + wave * (16 * QT);

    zero_pads<DT>(Ks, Vt, P.d);

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

    for (int kv0 = 0; kv0 < P.Nk; kv0 += KV_TILE) {
        __syncthreads();
        stage_rows<DT>(Ks, P.k, kbase, kv0, P.Nk, P.ldk, col0, P.d);
        stage_cols(Vt, P.v, kbase, kv0, P.Nk, P.ldv, col0, P.d);
        __syncthreads();

        f32x4 st[QT][4];
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) st[t][j] = fzero4();
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < DT; ++ks) {
                half4_t kf = ld4(Ks + (16 * j + c15) * RP + 16 * ks + 4 * g);
#pragma unroll
                for (int t = 0; t < QT; ++t) st[t][j] = mfma16(kf, qf[t][ks], st[t][j]);
            }
        half4_t pf[QT][4];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int kv = kv0 + 16 * j + 4 * g + i;
                    float s = kv < P.Nk ? st[t][j][i] * P.scale : -INFINITY;
                    st[t][j][i] = s;
                    mx = fmaxf(mx, s);
                }
            mx = grp_max(mx);
            const float mnew = fmaxf(m[t], mx);
            const float alpha = expf(m[t] - mnew);
            float rs = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float e = expf(st[t][j][i] - mnew);
                    rs += e;
                    pf[t][j][i] = (half_t)e;
                }
            rs = grp_sum(rs);
            l[t] = l[t] * alpha + rs;
            m[t] = mnew;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int i = 0; i < 4; ++i) oacc[t][dt][i] *= alpha;
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                half4_t vf = ld4(Vt + (16 * dt + c15) * TPAD + 16 * j + 4 * g);
#pragma unroll
                for (int t = 0; t < QT; ++t) oacc[t][dt] = mfma16(vf, pf[t][j], oacc[t][dt]);
            }
    }

#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = q0 + 16 * t + c15;
        if (qi >= P.Nq) continue;
        const float inv = 1.0f / l[t];
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
        if (g == 0 && lse) lse[((size_t)b * P.heads + h) * P.Nq + qi] = m[t] + logf(l[t]);
    }
}

// ---- backward: dQ (and D = rowsum(dO * O)) ---------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AParams P, const half_t* o, int ldo, const half_t* dO,
                                                           int lddo, const float* lse, float* Dbuf, half_t* dq,
                                                           int lddq) {
    constexpr int RP = DT * 16 + 8;
    MC_DYN_SMEM(smem);
    half_t* Ks = reinterpret_cast<half_t*>(smem);  // [64][RP]
    half_t* Vs = Ks + KV_TILE * RP;                // [64][RP]
    half_t* Kt = Vs + KV_TILE * RP;                // [DT*16][TPAD]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c15 = lane & 15;
    const int h = blockIdx.y, b = blockIdx.z;
    const int col0 = h * P.d;
    const size_t qbase = (size_t)b * P.Nq, kbase = (size_t)(b / P.kv_bdiv) * P.Nk;
    const int qi = blockIdx.x * 64 + wave * 16 + c15;
    const bool qok = qi < P.Nq;

    zero_pads<DT>(Ks, Kt, P.d);
    zero_pads<DT>(Vs, nullptr, P.d);

    half4_t qf[DT], dof[DT];
    float dsum = 0.f;
#pragma unroll
    for (int ks = 0; ks < DT; ++ks) {
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
    const float Dq = grp_sum(dsum);
    const size_t sidx = ((size_t)b * P.heads + h) * P.Nq + (qok ? qi : 0);
    const float lq = qok ? lse[sidx] : 0.f;
    if (qok && g == 0 && Dbuf) Dbuf[sidx] = Dq;

    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = fzero4();

    for (int kv0 = 0; kv0 < P.Nk; kv0 += KV_TILE) {
        __syncthreads();
        stage_rows<DT>(Ks, P.k, kbase, kv0, P.Nk, P.ldk, col0, P.d);
        stage_cols(Kt, P.k, kbase, kv0, P.Nk, P.ldk, col0, P.d);
        stage_rows<DT>(Vs, P.v, kbase, kv0, P.Nk, P.ldv, col0, P.d);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 sT = fzero4(), dpT = fzero4();
#pragma unroll
            for (int ks = 0; ks < DT; ++ks) {
                half4_t kf = ld4(Ks + (16 * j + c15) * RP + 16 * ks + 4 * g);
                half4_t vf = ld4(Vs + (16 * j + c15) * RP + 16 * ks + 4 * g);
                sT = mfma16(kf, qf[ks], sT);
                dpT = mfma16(vf, dof[ks], dpT);
            }
            half4_t dsf;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int kv = kv0 + 16 * j + 4 * g + i;
                float p = kv < P.Nk ? expf(sT[i] * P.scale - lq) : 0.f;
                dsf[i] = (half_t)(p * (dpT[i] - Dq));
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                half4_t kc = ld4(Kt + (16 * dt + c15) * TPAD + 16 * j + 4 * g);
                acc[dt] = mfma16(kc, dsf, acc[dt]);
            }
        }
    }
    if (!qok) return;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        int c = 16 * dt + 4 * g;
        if (c < P.d) {
            half4_t ov;
#pragma unroll
            for (int i = 0; i < 4; ++i) ov[i] = to_half(acc[dt][i] * P.scale);
            st4(dq + (qbase + qi) * lddq + col0 + c, ov);
        }
    }
}

// ---- backward: dK, dV (self-attention: kv batch == q batch) -----------------------------------
template <int DT>
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(AParams P, const half_t* dO, int lddo,
                                                             const float* lse, const float* Dbuf, half_t* dk,
                                                             int lddk, half_t* dv, int lddv) {
    constexpr int RP = DT * 16 + 8;
    MC_DYN_SMEM(smem);
    half_t* Qs = reinterpret_cast<half_t*>(smem);  // [64][RP]
    half_t* Os = Qs + KV_TILE * RP;                // [64][RP]   dO rows
    half_t* Qt = Os + KV_TILE * RP;                // [DT*16][TPAD]
    half_t* Ot = Qt + DT * 16 * TPAD;              // [DT*16][TPAD]
    float* lse_s = reinterpret_cast<float*>(Ot + DT * 16 * TPAD);  // [64]
    float* D_s = lse_s + KV_TILE;                                   // [64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c15 = lane & 15;
    const int h = blockIdx.y, b = blockIdx.z;
    const int col0 = h * P.d;
    const size_t qbase = (size_t)b * P.Nq, kbase = (size_t)b * P.Nk;
    const int kvi = blockIdx.x * 64 + wave * 16 + c15;
    const bool kok = kvi < P.Nk;

    zero_pads<DT>(Qs, Qt, P.d);
    zero_pads<DT>(Os, Ot, P.d);

    half4_t kf[DT], vf[DT];
#pragma unroll
    for (int ks = 0; ks < DT; ++ks) {
        int c = 16 * ks + 4 * g;
        if (kok && c < P.d) {
            kf[ks] = ld4(P.k + (kbase + kvi) * P.ldk + col0 + c);
            vf[ks] = ld4(P.v + (kbase + kvi) * P.ldv + col0 + c);
        } else {
            kf[ks] = zero4();
            vf[ks] = zero4();
        }
    }
    f32x4 ak[DT], av[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) ak[dt] = av[dt] = fzero4();

    for (int q0 = 0; q0 < P.Nq; q0 += KV_TILE) {
        __syncthreads();
        stage_rows<DT>(Qs, P.q, qbase, q0, P.Nq, P.ldq, col0, P.d);
        stage_cols(Qt, P.q, qbase, q0, P.Nq, P.ldq, col0, P.d);
        stage_rows<DT>(Os, dO, qbase, q0, P.Nq, lddo, col0, P.d);
        stage_cols(Ot, dO, qbase, q0, P.Nq, lddo, col0, P.d);
        if (threadIdx.x < KV_TILE) {
            int qi = q0 + threadIdx.x;
            size_t si = ((size_t)b * P.heads + h) * P.Nq + qi;
            lse_s[threadIdx.x] = qi < P.Nq ? lse[si] : INFINITY;
            D_s[threadIdx.x] = qi < P.Nq ? Dbuf[si] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 s = fzero4(), dp = fzero4();
#pragma unroll
            for (int ks = 0; ks < DT; ++ks) {
                half4_t qr = ld4(Qs + (16 * j + c15) * RP + 16 * ks + 4 * g);
                half4_t orr = ld4(Os + (16 * j + c15) * RP + 16 * ks + 4 * g);
                s = mfma16(qr, kf[ks], s);     // [q = 16j + 4g + i][kv = c15]
                dp = mfma16(orr, vf[ks], dp);
            }
            half4_t pf, dsf;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r = 16 * j + 4 * g + i;
                float p = expf(s[i] * P.scale - lse_s[r]);
                pf[i] = (half_t)p;
                dsf[i] = (half_t)(p * (dp[i] - D_s[r]));
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                half4_t oc = ld4(Ot + (16 * dt + c15) * TPAD + 16 * j + 4 * g);
                half4_t qc = ld4(Qt + (16 * dt + c15) * TPAD + 16 * j + 4 * g);
                av[dt] = mfma16(oc, pf, av[dt]);
                ak[dt] = mfma16(qc, dsf, ak[dt]);
            }
        }
    }
    if (!kok) return;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        int c = 16 * dt + 4 * g;
        if (c < P.d) {
            half4_t ok, ov;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ok[i] = to_half(ak[dt][i] * P.scale);
                ov[i] = to_half(av[dt][i]);
            }
            st4(dk + (kbase + kvi) * lddk + col0 + c, ok);
            st4(dv + (kbase + kvi) * lddv + col0 + c, ov);
        }
    }
}

static int a_check(const AParams& P) {
    if (P.Nq <= 0 || P.Nk <= 0 || P.heads <= 0 || P.d <= 0 || P.nbatch <= 0 || P.kv_bdiv <= 0) return 0;
    if (P.d % 8 || P.ldq % 8 || P.ldk % 8 || P.ldv % 8) return 0;
    return 1;
}

template <int DT>
static void a_launch_fwd(const AParams& P, half_t* o, int ldo, float* lse, hipStream_t s) {
    constexpr int RP = DT * 16 + 8;
    size_t smem = (size_t)(KV_TILE * RP + DT * 16 * TPAD) * sizeof(half_t);
    if (P.Nq >= 512) {
        dim3 grid((P.Nq + 127) / 128, P.heads, P.nbatch);
        MC_LAUNCH((attn_fwd_kernel<DT, 2>), grid, dim3(256), smem, s, P, o, ldo, lse);
    } else {
        dim3 grid((P.Nq + 63) / 64, P.heads, P.nbatch);
        MC_LAUNCH((attn_fwd_kernel<DT, 1>), grid, dim3(256), smem, s, P, o, ldo, lse);
    }
}
template <int DT>
static void a_launch_dq(const AParams& P, const half_t* o, int ldo, const half_t* dO, int lddo, const float* lse,
                        float* Dbuf, half_t* dq, int lddq, hipStream_t s) {
    constexpr int RP = DT * 16 + 8;
    size_t smem = (size_t)(2 * KV_TILE * RP + DT * 16 * TPAD) * sizeof(half_t);
    dim3 grid((P.Nq + 63) / 64, P.heads, P.nbatch);
    allow_big_smem(attn_bwd_dq_kernel<DT>, smem);
    MC_LAUNCH((attn_bwd_dq_kernel<DT>), grid, dim3(256), smem, s, P, o, ldo, dO, lddo, lse, Dbuf, dq, lddq);
}
template <int DT>
static void a_launch_dkdv(const AParams& P, const half_t* dO, int lddo, const float* lse, const float* Dbuf,
                          half_t* dk, int lddk, half_t* dv, int lddv, hipStream_t s) {
    constexpr int RP = DT * 16 + 8;
    size_t smem = (size_t)(2 * KV_TILE * RP + 2 * DT * 16 * TPAD) * sizeof(half_t) + 2 * KV_TILE * sizeof(float);
    dim3 grid((P.Nk + 63) / 64, P.heads, P.nbatch);
    allow_big_smem(attn_bwd_dkdv_kernel<DT>, smem);
    MC_LAUNCH((attn_bwd_dkdv_kernel<DT>), grid, dim3(256), smem, s, P, dO, lddo, lse, Dbuf, dk, lddk, dv, lddv);
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
    P.nbatch = nbatch; P.kv_bdiv = kv_bdiv; P.scale = scale;
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
#define CALL(DT_) a_launch_fwd<DT_>(P, (half_t*)o, ldo, lse, s)
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
#define CALL(DT_)                                                                                             \
    a_launch_dq<DT_>(P, (const half_t*)o, ldo, (const half_t*)dO, lddo, lse, Dbuf, (half_t*)dq, lddq, s);     \
    if (dk) a_launch_dkdv<DT_>(P, (const half_t*)dO, lddo, lse, Dbuf, (half_t*)dk, lddk, (half_t*)dv, lddv, s)
    MC_A_DISPATCH(CALL)
#undef CALL
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}
