// Shared GEMM epilogue: rows of an fp32 LDS staging tile -> bias / residual / fused GEGLU -> coalesced 16-byte
// stores.  A thread owns one 8-column segment (fixed) and walks NIT rows.
//
// Latency shape matters here: with K = 320 the main loop of a 256x320 tile is only 5 k-steps, so the epilogue is a
// large share of the kernel.  Every global read it needs is therefore issued before the data is wanted:
//   * the bias segment is loaded once per thread (a tile almost never straddles two batch entries; if it does the
//     per-row path is taken),
//   * the residual rows of a pass are fetched into registers BEFORE the accumulators go to LDS and the barrier,
//     so their HBM latency overlaps the staging instead of serialising row after row behind it.
#pragma once
#include "gemm_params.hpp"

namespace mc {

template <int TPR, int RPP, int NIT, int CS>
struct Epilogue {
    int col, n, nvalid, rsub;
    bool active, vec16, bias_rowwise;
    float bv[8];
    half8_t rres[NIT];

    __device__ __forceinline__ void init(const GemmParams& p, int tid, int n0, int m_first, int m_last) {
        col = (tid % TPR) * 8;
        rsub = tid / TPR;
        n = n0 + col;
        active = tid < RPP * TPR && n < p.N;
        nvalid = min(8, p.N - n);  // multiple of 4
        vec16 = !(p.N & 7) && !(p.ldc & 7) && (!p.R || !(p.ldr & 7));
        bias_rowwise = false;
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = 0.f;
        if (p.bias && active) {
            const int b0 = m_first / p.rows_per_batch, b1 = m_last / p.rows_per_batch;
            if (b0 == b1)
                load_bias(p.bias + (size_t)b0 * p.N + n);
            else
                bias_rowwise = true;
        }
    }
    __device__ __forceinline__ void load_bias(const float* brow) {
        f32x4 x = *reinterpret_cast<const f32x4*>(brow);
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = x[e];
        if (nvalid == 8) {
            f32x4 y = *reinterpret_cast<const f32x4*>(brow + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[4 + e] = y[e];
        }
    }
    // staging rows [0, rows) hold global rows m_base + row
    __device__ __forceinline__ void prefetch(const GemmParams& p, int m_base, int rows) {
        if (!(active && p.R && vec16 && p.epi != 1)) return;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int rl = it * RPP + rsub, m = m_base + rl;
            if (rl < rows && m < p.M) rres[it] = ld8(p.R + (size_t)m * p.ldr + n);
        }
    }
    // split-K: raw fp32 partial sums of this K range -> ws[split][m][n] (two 16-byte stores per segment)
    __device__ __forceinline__ void store_partial(const GemmParams& p, const float* Cs, int m_base, int rows, int split) {
        if (!active) return;
        float* ws = p.ws + (size_t)split * p.M * p.N;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int rl = it * RPP + rsub, m = m_base + rl;
            if (rl >= rows || m >= p.M) continue;
            float* dst = ws + (size_t)m * p.N + n;
            *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(Cs + rl * CS + col);
            if (nvalid == 8) *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(Cs + rl * CS + col + 4);
        }
    }
    __device__ __forceinline__ void store(const GemmParams& p, const float* Cs, int m_base, int rows) {
        if (!active) return;
        if (vec16 && p.epi == 0 && !bias_rowwise) {
            // the common case, kept free of per-row branching: 16-byte segments, bias already in registers
            const bool has_bias = p.bias != nullptr, has_res = p.R != nullptr;
            half_t* crow = p.C + (size_t)(m_base + rsub) * p.ldc + n;
            const float* srow = Cs + rsub * CS + col;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int rl = it * RPP + rsub;
                if (rl < rows && m_base + rl < p.M) {
                    f32x4 a = *reinterpret_cast<const f32x4*>(srow + it * RPP * CS);
                    f32x4 b = *reinterpret_cast<const f32x4*>(srow + it * RPP * CS + 4);
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = a[e];
                        v[4 + e] = b[e];
                    }
                    if (has_bias) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += bv[e];
                    }
                    if (has_res) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)rres[it][e];
                    }
                    half8_t o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = to_half(v[e]);
                    if (p.dbg & 32) {   // experiment (MC_GEMM_DEBUG=32): streaming (nt) stores for the output tile
#ifndef MC_EMU
                        __builtin_nontemporal_store(o, reinterpret_cast<half8_t*>(crow + (size_t)it * RPP * p.ldc));
#endif
                    } else if (!(p.dbg & 1)) st8(crow + (size_t)it * RPP * p.ldc, o);
                }
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int rl = it * RPP + rsub, m = m_base + rl;
            if (rl >= rows || m >= p.M) continue;
            float v[8];
            {
                f32x4 a = *reinterpret_cast<const f32x4*>(Cs + rl * CS + col);
                f32x4 b = *reinterpret_cast<const f32x4*>(Cs + rl * CS + col + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = a[e];
                    v[4 + e] = b[e];
                }
            }
            if (bias_rowwise) load_bias(p.bias + (size_t)(m / p.rows_per_batch) * p.N + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bv[e];
            if (p.epi == 1) {
                // fused GEGLU: weight rows are interleaved (h_j, gate_j); out[m][n/2 + j] = h_j * gelu(gate_j)
                half4_t o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = to_half(v[2 * e] * gelu_f(v[2 * e + 1]));
                half_t* dst = p.C + (size_t)m * p.ldc + (n >> 1);
                if (p.dbg & 1) continue;
                if (nvalid == 8) {
                    st4(dst, o);
                } else {
                    dst[0] = o[0];
                    dst[1] = o[1];
                }
                continue;
            }
            if (vec16) {
                if (p.R) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rres[it][e];
                }
                half8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = to_half(v[e]);
                if (!(p.dbg & 1)) st8(p.C + (size_t)m * p.ldc + n, o);
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (4 * h >= nvalid) break;
                    half4_t o;
                    if (p.R) {
                        half4_t r = ld4(p.R + (size_t)m * p.ldr + n + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * h + e] += (float)r[e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = to_half(v[4 * h + e]);
                    st4(p.C + (size_t)m * p.ldc + n + 4 * h, o);
                }
            }
        }
    }
};

}  // namespace mc
