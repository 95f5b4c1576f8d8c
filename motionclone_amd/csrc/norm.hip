// GroupNorm(32 groups)[+SiLU] and LayerNorm, forward and data-gradient, on
// channels-last [frames, HW, C] fp16 activations.
// Reference semantics: resnet.py:21-29,186-187,197-203 (eps 1e-5, +SiLU),
// attention.py:61,105 / motion_module.py:112,145 (eps 1e-6), LayerNorm at
// attention.py:189,206,212 and motion_module.py:204,210; the sinusoidal temporal
// position table (motion_module.py:237-246) is added in the LayerNorm epilogue.
//
// All of these are HBM-bound: 16 B per lane loads, fp32 statistics, the skip
// concat of the up blocks is read as two sources instead of being materialised.
#include "mc_common.hpp"
#include "gn_stats.hpp"

namespace mc {

struct GnSrc {
    const half_t* a;
    const half_t* b;
    int lda, ldb;
    int c1, ctot;  // channels from a; total
    int hw;        // tokens per frame
    int cpg;       // channels per group (ctot / 32)
};

__device__ __forceinline__ half8_t gn_load(const GnSrc& s, size_t tok, int c) {
    return c < s.c1 ? ld8(s.a + tok * s.lda + c) : ld8(s.b + tok * s.ldb + (c - s.c1));
}

// Group totals of two [R][ctot] per-channel partial arrays in LDS -> out[g] = (sum, sum) for the 32 groups.  SUB adjacent
// lanes share a group (32 * SUB <= blockDim.x, SUB a power of two), each sums every SUB-th of the R * cpg partials, then a
// log2(SUB)-step butterfly; fixed summation order (deterministic).  Replaces a 32-thread serial walk that was the tail of
// every block (R * cpg * 2 dependent LDS reads).
__device__ __forceinline__ void gn_group_totals(const float* p1, const float* p2, int R, int ctot, int cpg, float* out) {
    const int t = threadIdx.x;
    int SUB = 1;
    while (64 * SUB <= (int)blockDim.x && SUB < 8) SUB *= 2;
    const int gsel = t / SUB, sub = t % SUB;
    float a = 0.f, b = 0.f;
    if (gsel < 32) {
        const int items = R * cpg;
        for (int it = sub; it < items; it += SUB) {
            const int rr = it / cpg, c = gsel * cpg + it % cpg;
            a += p1[rr * ctot + c];
            b += p2[rr * ctot + c];
        }
    }
    for (int mk = 1; mk < SUB; mk <<= 1) {   // every lane of the wave takes part (lanes with gsel >= 32 carry zeros)
        a += shfl_xor(a, mk);
        b += shfl_xor(b, mk);
    }
    if (gsel < 32 && sub == 0) {
        out[gsel * 2] = a;
        out[gsel * 2 + 1] = b;
    }
}

// ---- pass 1 of forward stats: per (frame, chunk, group) sum / sum of squares -------------
// grid (nchunk, frames), block = VC * R threads (VC = ctot/8 vector columns)
__global__ void gn_partial_kernel(GnSrc s, int R, int nchunk, float* partial) {
    MC_DYN_SMEM(smem);
    float* ssum = reinterpret_cast<float*>(smem);  // [R][ctot]
    float* ssq = ssum + R * s.ctot;
    const int VC = s.ctot / 8;
    const int t = threadIdx.x;
    const int col = t % VC, r = t / VC;
    const int frame = blockIdx.y, chunk = blockIdx.x;
    const int per = (s.hw + nchunk - 1) / nchunk;
    const int t0 = chunk * per;
    const int t1 = min(s.hw, t0 + per);
    float sum[8], sq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sum[e] = sq[e] = 0.f;
    if (r < R) {
        // four independent 16-byte loads in flight per thread (the loop is a pure HBM stream)
        int tk = t0 + r;
        for (; tk + 3 * R < t1; tk += 4 * R) {
            half8_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = gn_load(s, (size_t)frame * s.hw + tk + u * R, col * 8);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = (float)v[u][e];
                    sum[e] += f;
                    sq[e] += f * f;
                }
        }
        for (; tk < t1; tk += R) {
            half8_t v = gn_load(s, (size_t)frame * s.hw + tk, col * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = (float)v[e];
                sum[e] += f;
                sq[e] += f * f;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ssum[r * s.ctot + col * 8 + e] = sum[e];
            ssq[r * s.ctot + col * 8 + e] = sq[e];
        }
    }
    __syncthreads();
    gn_group_totals(ssum, ssq, R, s.ctot, s.cpg, partial + (size_t)(frame * nchunk + chunk) * 64);
}

// mode 0: (sum, sumsq) -> (mean, rstd);  mode 1: (s1, s2) -> (s1/n, s2/n)
__global__ void gn_finalize_kernel(const float* partial, float* stats, int frames, int nchunk, float n,
                                   float eps, int mode) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= frames * 32) return;
    int frame = i / 32, g = i % 32;
    float a = 0.f, b = 0.f;
    for (int c = 0; c < nchunk; ++c) {
#pragma clang fp reassociate(off)   // chunk order 0 .. nchunk-1 exactly (fast-math would split the reduction): gn_block_stats adds alike
        const float* q = partial + ((size_t)(frame * nchunk + c) * 32 + g) * 2;
        a += q[0];
        b += q[1];
    }
    if (mode == 0) {
        float mean = a / n;
        float var = fmaxf(b / n - mean * mean, 0.f);
        stats[i * 2] = mean;
        stats[i * 2 + 1] = 1.0f / sqrtf(var + eps);
    } else {
        stats[i * 2] = a / n;
        stats[i * 2 + 1] = b / n;
    }
}

// y = (x - mean) * rstd * gamma + beta, optional SiLU; out is [tokens][ctot] (ld = ldo).
// grid (nchunk, frames), block = VC * R threads: a thread owns one 8-channel vector column for the whole chunk, so
// the per-channel scale / shift (rstd*gamma, beta - mean*rstd*gamma) are computed once and live in registers.
// partial != nullptr: the statistics come from the per-chunk partials (gn_block_stats) and `stats` is an OUTPUT; `pchunks` = the
// chunks per frame those partials were written with (gn_partial_kernel: nchunk; a producing GEMM's epilogue: hw / 64 or hw / 32).
__global__ void gn_apply_kernel(GnSrc s, float* stats, const float* gamma, const float* beta,
                                half_t* out, int ldo, int R, int nchunk, int silu, const float* partial, float n, float eps,
                                int pchunks) {
    __shared__ float bst[64];
    const int VC = s.ctot / 8;
    const int t = threadIdx.x;
    const int col = t % VC, r = t / VC;
    const int frame = blockIdx.y, chunk = blockIdx.x;
    if (partial) gn_block_stats(partial, frame, pchunks, n, eps, 0, bst, stats, chunk == 0);
    if (r >= R) return;
    const int per = (s.hw + nchunk - 1) / nchunk;
    const int t0 = chunk * per;
    const int t1 = min(s.hw, t0 + per);
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int c = col * 8 + e;
        const float* st = partial ? bst + (c / s.cpg) * 2 : stats + ((size_t)frame * 32 + c / s.cpg) * 2;
        sc[e] = st[1] * gamma[c];
        sh[e] = beta[c] - st[0] * sc[e];
    }
    for (int tk = t0 + r; tk < t1; tk += R) {
        size_t tok = (size_t)frame * s.hw + tk;
        half8_t v = gn_load(s, tok, col * 8);
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = (float)v[e] * sc[e] + sh[e];
            if (silu) y = silu_f(y);
            o[e] = to_half(y);
        }
        st8(out + tok * ldo + col * 8, o);
    }
}

// ---- backward pass 1: per (frame, chunk, group) sums of dxhat and dxhat*xhat ---------------
__global__ void gn_bwd_partial_kernel(GnSrc s, const half_t* dz, int lddz, const float* stats,
                                      const float* gamma, const float* beta, int silu, int R, int nchunk,
                                      float* partial) {
    MC_DYN_SMEM(smem);
    float* s1 = reinterpret_cast<float*>(smem);
    float* s2 = s1 + R * s.ctot;
    const int VC = s.ctot / 8;
    const int t = threadIdx.x;
    const int col = t % VC, r = t / VC;
    const int frame = blockIdx.y, chunk = blockIdx.x;
    const int per = (s.hw + nchunk - 1) / nchunk;
    const int t0 = chunk * per;
    const int t1 = min(s.hw, t0 + per);
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = b[e] = 0.f;
    if (r < R) {
        float mean[8], rstd[8], gm[8], bt[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int c = col * 8 + e;
            const float* st = stats + ((size_t)frame * 32 + c / s.cpg) * 2;
            mean[e] = st[0];
            rstd[e] = st[1];
            gm[e] = gamma[c];
            bt[e] = beta[c];
        }
        for (int tk = t0 + r; tk < t1; tk += R) {
            size_t tok = (size_t)frame * s.hw + tk;
            half8_t x = gn_load(s, tok, col * 8);
            half8_t g = ld8(dz + tok * lddz + col * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float xh = ((float)x[e] - mean[e]) * rstd[e];
                float dy = (float)g[e];
                if (silu) dy *= silu_grad_f(xh * gm[e] + bt[e]);
                float dxh = dy * gm[e];
                a[e] += dxh;
                b[e] += dxh * xh;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s1[r * s.ctot + col * 8 + e] = a[e];
            s2[r * s.ctot + col * 8 + e] = b[e];
        }
    }
    __syncthreads();
    gn_group_totals(s1, s2, R, s.ctot, s.cpg, partial + (size_t)(frame * nchunk + chunk) * 64);
}

// dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat*xhat)); optional accumulate into dx.  Same thread layout
// as gn_apply_kernel (per-channel constants in registers).
// (the group means of dxhat and dxhat * xhat come from the per-chunk partials of pass 1: gn_block_stats, mode 1)
__global__ void gn_bwd_apply_kernel(GnSrc s, const half_t* dz, int lddz, const float* stats,
                                    float* bstats_out, const float* gamma, const float* beta, int silu,
                                    half_t* dx, int lddx, int R, int nchunk, int accumulate, const float* partial, float n) {
    __shared__ float bstats[64];
    const int VC = s.ctot / 8;
    const int t = threadIdx.x;
    const int col = t % VC, r = t / VC;
    const int frame = blockIdx.y, chunk = blockIdx.x;
    gn_block_stats(partial, frame, nchunk, n, 0.f, 1, bstats, bstats_out, chunk == 0);
    if (r >= R) return;
    const int per = (s.hw + nchunk - 1) / nchunk;
    const int t0 = chunk * per;
    const int t1 = min(s.hw, t0 + per);
    float mean[8], rstd[8], gm[8], bt[8], m1[8], m2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int c = col * 8 + e;
        size_t gi = ((size_t)frame * 32 + c / s.cpg) * 2;
        mean[e] = stats[gi];
        rstd[e] = stats[gi + 1];
        m1[e] = bstats[(c / s.cpg) * 2];
        m2[e] = bstats[(c / s.cpg) * 2 + 1];
        gm[e] = gamma[c];
        bt[e] = beta[c];
    }
    for (int tk = t0 + r; tk < t1; tk += R) {
        size_t tok = (size_t)frame * s.hw + tk;
        half8_t x = gn_load(s, tok, col * 8);
        half8_t g = ld8(dz + tok * lddz + col * 8);
        half8_t prev;
        if (accumulate) prev = ld8(dx + tok * lddx + col * 8);
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float xh = ((float)x[e] - mean[e]) * rstd[e];
            float dy = (float)g[e];
            if (silu) dy *= silu_grad_f(xh * gm[e] + bt[e]);
            float v = rstd[e] * (dy * gm[e] - m1[e] - xh * m2[e]);
            if (accumulate) v += (float)prev[e];
            o[e] = to_half(v);
        }
        st8(dx + tok * lddx + col * 8, o);
    }
}

// ---- LayerNorm: one wave per row, row kept in registers ----------------------------------
// A wave walks LN_RPW consecutive rows (fully unrolled, so the loads of all of them are in flight together).
// gamma / beta of the lane's 8-channel vectors are loaded once per wave as 16-byte vectors.
constexpr int LN_RPW = 4;

__device__ __forceinline__ void ld8f(const float* p, float (&d)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        d[e] = a[e];
        d[4 + e] = b[e];
    }
}

template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const half_t* x, int ldx, half_t* y, int ldy,
                                                      const float* gamma, const float* beta,
                                                      const float* pe, int hw, int nframes_pe,
                                                      float* stats, int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * LN_RPW;
    const int nvec = C / 8;
    float gam[NV][8], bet[NV][8];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int vi = lane + 64 * i;
        if (vi < nvec) {
            ld8f(gamma + vi * 8, gam[i]);
            ld8f(beta + vi * 8, bet[i]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) gam[i][e] = bet[i][e] = 0.f;
        }
    }
    half8_t v[LN_RPW][NV];
    float mean[LN_RPW], rstd[LN_RPW];
#pragma unroll
    for (int rr = 0; rr < LN_RPW; ++rr) {
        const int row = row0 + rr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int vi = lane + 64 * i;
            v[rr][i] = (row < M && vi < nvec) ? ld8(x + (size_t)row * ldx + vi * 8) : zero8();
        }
    }
#pragma unroll
    for (int rr = 0; rr < LN_RPW; ++rr) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += (float)v[rr][i][e];
        mean[rr] = wave_sum(sum) / C;
    }
#pragma unroll
    for (int rr = 0; rr < LN_RPW; ++rr) {
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int vi = lane + 64 * i;
            if (vi < nvec) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float d = (float)v[rr][i][e] - mean[rr];
                    sq += d * d;
                }
            }
        }
        rstd[rr] = 1.0f / sqrtf(wave_sum(sq) / C + eps);
    }
#pragma unroll
    for (int rr = 0; rr < LN_RPW; ++rr) {
        const int row = row0 + rr;
        if (row >= M) continue;
        if (lane == 0 && stats) {
            stats[(size_t)row * 2] = mean[rr];
            stats[(size_t)row * 2 + 1] = rstd[rr];
        }
        const float* perow = pe ? pe + (size_t)((row / hw) % nframes_pe) * C : nullptr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int vi = lane + 64 * i;
            if (vi < nvec) {
                float pv[8];
                if (perow) {
                    ld8f(perow + vi * 8, pv);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) pv[e] = 0.f;
                }
                half8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    o[e] = to_half(((float)v[rr][i][e] - mean[rr]) * rstd[rr] * gam[i][e] + bet[i][e] + pv[e]);
                st8(y + (size_t)row * ldy + vi * 8, o);
            }
        }
    }
}

// LayerNorm forward for the widths of the UNet (C = 320 / 640 / 1280 = 40 vectors x LPR lanes): a row is owned by LPR
// adjacent lanes (8 / 16 / 32), five 16-byte vectors each, so all 64 lanes of a wave carry data (the one-row-per-wave
// kernel above leaves 24 of 64 lanes idle at C = 320) and a wave streams 64 / LPR rows at once; row statistics by an
// LPR-lane butterfly.  Same arithmetic (two-pass variance), same outputs.
template <int LPR>
__global__ __launch_bounds__(256) void ln_fwd5_kernel(const half_t* x, int ldx, half_t* y, int ldy, const float* gamma,
                                                       const float* beta, const float* pe, int hw, int nframes_pe,
                                                       float* stats, int M, int C, float eps) {
    constexpr int RW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int j = lane % LPR;
    const int row = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * RW + lane / LPR;
    const bool live = row < M;
    half8_t v[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) v[i] = live ? ld8(x + (size_t)row * ldx + (j + LPR * i) * 8) : zero8();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += (float)v[i][e];
#pragma unroll
    for (int mk = 1; mk < LPR; mk <<= 1) sum += shfl_xor(sum, mk);
    const float mean = sum / C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float d = (float)v[i][e] - mean;
            sq += d * d;
        }
#pragma unroll
    for (int mk = 1; mk < LPR; mk <<= 1) sq += shfl_xor(sq, mk);
    const float rstd = 1.0f / sqrtf(sq / C + eps);
    if (!live) return;
    if (j == 0 && stats) {
        stats[(size_t)row * 2] = mean;
        stats[(size_t)row * 2 + 1] = rstd;
    }
    const float* perow = pe ? pe + (size_t)((row / hw) % nframes_pe) * C : nullptr;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int c0 = (j + LPR * i) * 8;
        float gm[8], bt[8], pv[8];
        ld8f(gamma + c0, gm);
        ld8f(beta + c0, bt);
        if (perow) {
            ld8f(perow + c0, pv);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = 0.f;
        }
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = to_half(((float)v[i][e] - mean) * rstd * gm[e] + bt[e] + pv[e]);
        st8(y + (size_t)row * ldy + c0, o);
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)) [+ add], g = dy * gamma
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const half_t* dy, int lddy, const half_t* x, int ldx,
                                                      const float* stats, const float* gamma,
                                                      const half_t* add, int ldadd, half_t* dx, int lddx,
                                                      int M, int C) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * LN_RPW;
    const int nvec = C / 8;
    float gam[NV][8];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int vi = lane + 64 * i;
        if (vi < nvec) {
            ld8f(gamma + vi * 8, gam[i]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) gam[i][e] = 0.f;
        }
    }
#pragma unroll 2
    for (int rr = 0; rr < LN_RPW; ++rr) {
        const int row = row0 + rr;
        const bool live = row < M;
        const float mean = live ? stats[(size_t)row * 2] : 0.f;
        const float rstd = live ? stats[(size_t)row * 2 + 1] : 0.f;
        float g[NV][8], xh[NV][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int vi = lane + 64 * i;
            if (live && vi < nvec) {
                half8_t xv = ld8(x + (size_t)row * ldx + vi * 8);
                half8_t dv = ld8(dy + (size_t)row * lddy + vi * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[i][e] = ((float)xv[e] - mean) * rstd;
                    g[i][e] = (float)dv[e] * gam[i][e];
                    s1 += g[i][e];
                    s2 += g[i][e] * xh[i][e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) xh[i][e] = g[i][e] = 0.f;
            }
        }
        s1 = wave_sum(s1) / C;
        s2 = wave_sum(s2) / C;
        if (!live) continue;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int vi = lane + 64 * i;
            if (vi < nvec) {
                half8_t o;
                half8_t av;
                if (add) av = ld8(add + (size_t)row * ldadd + vi * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t = rstd * (g[i][e] - s1 - xh[i][e] * s2);
                    if (add) t += (float)av[e];
                    o[e] = to_half(t);
                }
                st8(dx + (size_t)row * lddx + vi * 8, o);
            }
        }
    }
}

static int gn_geometry(int ctot, int* R, int* threads) {
    int VC = ctot / 8;
    if (VC > 1024) return 0;
    int r = 256 / VC;
    if (r < 1) r = 1;
    *R = r;
    *threads = ((VC * r + 63) / 64) * 64;
    if (*threads < 64) *threads = 64;
    return 1;
}

}  // namespace mc

using namespace mc;

static int make_src(GnSrc* s, const void* a, const void* b, int lda, int ldb, int c1, int ctot, int hw) {
    if (ctot <= 0 || ctot % 32 || ctot % 8 || c1 % 8 || c1 > ctot || c1 <= 0) return 0;
    if (c1 < ctot && !b) return 0;
    if (lda % 8 || (b && ldb % 8)) return 0;
    s->a = (const half_t*)a; s->b = (const half_t*)b; s->lda = lda; s->ldb = ldb;
    s->c1 = c1; s->ctot = ctot; s->hw = hw; s->cpg = ctot / 32;
    return 1;
}

// chunks per frame of the two-stage GroupNorm kernels (grid = chunks x frames).  64-row chunks at the large levels; the
// small levels (16x16, 8x8 latents) get shorter chunks so that 32 frames still put >= 256 workgroups on the chip:
// with 64-row chunks a 16x16-level apply ran on 128 workgroups at 1.2 TB/s
extern "C" int mc_gn_nchunk(int hw) {
    const int rows = hw >= 4096 ? 64 : (hw >= 1024 ? 32 : (hw >= 256 ? 16 : 8));
    int n = (hw + rows - 1) / rows;
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    return n;
}

// bytes of the `partial` workspace of mc_groupnorm_stats_f16 / mc_groupnorm_bwd_f16, and of the latter's `bstats`
extern "C" long mc_workspace_bytes_groupnorm(int frames, int hw) {
    return (frames <= 0 || hw <= 0) ? -1 : (long)sizeof(float) * frames * mc_gn_nchunk(hw) * 64;
}
extern "C" long mc_workspace_bytes_groupnorm_bwd_stats(int frames) {
    return frames <= 0 ? -1 : (long)sizeof(float) * frames * 64;
}

namespace mc {
// pass 1 alone (mc_norm_gemm_f16, kind 2: the statistics are finalised inside the GEMM that applies them)
int gn_partial_launch(const void* a, int lda, int ctot, int frames, int hw, float* partial, hipStream_t stream) {
    GnSrc s;
    if (!make_src(&s, a, nullptr, lda, 0, ctot, ctot, hw) || frames <= 0 || hw <= 0) return MC_ERR_SHAPE;
    int R, threads;
    if (!gn_geometry(ctot, &R, &threads)) return MC_ERR_UNSUPPORTED;
    const int nchunk = mc_gn_nchunk(hw);
    MC_LAUNCH(gn_partial_kernel, dim3(nchunk, frames), dim3(threads), (size_t)2 * R * ctot * sizeof(float), stream, s, R, nchunk,
              partial);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}
}  // namespace mc

// workspace: float partial[frames * nchunk * 32 * 2]; output stats: float[frames*32*2] (mean, rstd)
extern "C" int mc_groupnorm_stats_f16(const void* a, const void* b, int lda, int ldb, int c1, int ctot,
                                      int frames, int hw, float eps, float* partial, float* stats,
                                      void* stream) {
    GnSrc s;
    if (!make_src(&s, a, b, lda, ldb, c1, ctot, hw) || frames <= 0 || hw <= 0) return MC_ERR_SHAPE;
    int R, threads;
    if (!gn_geometry(ctot, &R, &threads)) return MC_ERR_UNSUPPORTED;
    int nchunk = mc_gn_nchunk(hw);
    size_t smem = (size_t)2 * R * ctot * sizeof(float);
    MC_LAUNCH(gn_partial_kernel, dim3(nchunk, frames), dim3(threads), smem, (hipStream_t)stream, s, R, nchunk,
              partial);
    int n = frames * 32;
    MC_LAUNCH(gn_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
              (const float*)partial, stats, frames, nchunk, (float)hw * s.cpg, eps, 0);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

extern "C" int mc_groupnorm_apply_f16(const void* a, const void* b, int lda, int ldb, int c1, int ctot,
                                      int frames, int hw, const float* stats, const float* gamma,
                                      const float* beta, void* out, int ldo, int silu, void* stream) {
    GnSrc s;
    if (!make_src(&s, a, b, lda, ldb, c1, ctot, hw) || ldo % 8) return MC_ERR_SHAPE;
    int R, threads;
    if (!gn_geometry(ctot, &R, &threads)) return MC_ERR_UNSUPPORTED;
    int nchunk = mc_gn_nchunk(hw);
    MC_LAUNCH(gn_apply_kernel, dim3(nchunk, frames), dim3(threads), 0, (hipStream_t)stream, s, const_cast<float*>(stats), gamma,
              beta, (half_t*)out, ldo, R, nchunk, silu, (const float*)nullptr, 0.f, 0.f, 0);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

// statistics + normalisation (+ SiLU) of one GroupNorm in TWO launches: per-chunk partial sums, then the apply pass, which
// finalises the statistics in its prologue and writes them to `stats` (float[frames*32*2], for the backward).  Same sums in the
// same order as mc_groupnorm_stats_f16 followed by mc_groupnorm_apply_f16; (mean, rstd) are the same expressions compiled in
// two kernels under fast-math, so they may differ by an fp32 rounding and the output by one fp16 step on a few elements
// (tests/test_kernels.py::test_groupnorm_fwd_bwd bounds both).  workspace: partial as for mc_groupnorm_stats_f16.
extern "C" int mc_groupnorm_fwd_f16(const void* a, const void* b, int lda, int ldb, int c1, int ctot, int frames, int hw,
                                    float eps, float* partial, float* stats, const float* gamma, const float* beta,
                                    void* out, int ldo, int silu, void* stream) {
    GnSrc s;
    if (!make_src(&s, a, b, lda, ldb, c1, ctot, hw) || frames <= 0 || hw <= 0 || ldo % 8) return MC_ERR_SHAPE;
    int R, threads;
    if (!gn_geometry(ctot, &R, &threads) || threads < 64) return MC_ERR_UNSUPPORTED;
    int nchunk = mc_gn_nchunk(hw);
    size_t smem = (size_t)2 * R * ctot * sizeof(float);
    MC_LAUNCH(gn_partial_kernel, dim3(nchunk, frames), dim3(threads), smem, (hipStream_t)stream, s, R, nchunk, partial);
    MC_LAUNCH(gn_apply_kernel, dim3(nchunk, frames), dim3(threads), 0, (hipStream_t)stream, s, stats, gamma, beta,
              (half_t*)out, ldo, R, nchunk, silu, (const float*)partial, (float)hw * s.cpg, eps, nchunk);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

// mc_groupnorm_fwd_f16 WITHOUT its statistics pass (round 6): the per-chunk partial sums were left by the kernel that produced
// `a` (mc_gemm_gnstats_f16: `pchunks` = hw / the chunk height it returned).  ONE launch; single source.  The chunk sums are added
// in chunk order by every workgroup's prologue (gn_block_stats), as for mc_groupnorm_fwd_f16.
extern "C" int mc_groupnorm_fwd_partial_f16(const void* a, int lda, int ctot, int frames, int hw, float eps,
                                            const float* partial, int pchunks, float* stats, const float* gamma,
                                            const float* beta, void* out, int ldo, int silu, void* stream) {
    GnSrc s;
    if (!make_src(&s, a, nullptr, lda, 0, ctot, ctot, hw) || frames <= 0 || hw <= 0 || ldo % 8) return MC_ERR_SHAPE;
    if (!partial || pchunks <= 0 || pchunks > 256) return MC_ERR_SHAPE;
    int R, threads;
    if (!gn_geometry(ctot, &R, &threads) || threads < 64) return MC_ERR_UNSUPPORTED;
    int nchunk = mc_gn_nchunk(hw);
    MC_LAUNCH(gn_apply_kernel, dim3(nchunk, frames), dim3(threads), 0, (hipStream_t)stream, s, stats, gamma, beta,
              (half_t*)out, ldo, R, nchunk, silu, partial, (float)hw * s.cpg, eps, pchunks);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

// workspace: partial as above + bstats float[frames*32*2]
extern "C" int mc_groupnorm_bwd_f16(const void* a, const void* b, int lda, int ldb, int c1, int ctot,
                                    int frames, int hw, const void* dz, int lddz, const float* stats,
                                    const float* gamma, const float* beta, int silu, float* partial,
                                    float* bstats, void* dx, int lddx, int accumulate, void* stream) {
    GnSrc s;
    if (!make_src(&s, a, b, lda, ldb, c1, ctot, hw) || lddz % 8 || lddx % 8) return MC_ERR_SHAPE;
    int R, threads;
    if (!gn_geometry(ctot, &R, &threads) || threads < 64) return MC_ERR_UNSUPPORTED;   // checked BEFORE anything is launched
    int nchunk = mc_gn_nchunk(hw);
    size_t smem = (size_t)2 * R * ctot * sizeof(float);
    MC_LAUNCH(gn_bwd_partial_kernel, dim3(nchunk, frames), dim3(threads), smem, (hipStream_t)stream, s,
              (const half_t*)dz, lddz, stats, gamma, beta, silu, R, nchunk, partial);
    MC_LAUNCH(gn_bwd_apply_kernel, dim3(nchunk, frames), dim3(threads), 0, (hipStream_t)stream, s, (const half_t*)dz,
              lddz, stats, bstats, gamma, beta, silu, (half_t*)dx, lddx, R, nchunk, accumulate, (const float*)partial,
              (float)hw * s.cpg);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

extern "C" int mc_layernorm_fwd_f16(const void* x, int ldx, void* y, int ldy, const float* gamma,
                                    const float* beta, const float* pe, int hw, int nframes_pe,
                                    float* stats, int M, int C, float eps, void* stream) {
    if (M <= 0 || C <= 0 || C % 8 || ldx % 8 || ldy % 8 || C > 1536) return MC_ERR_SHAPE;
    if (pe && (hw <= 0 || nframes_pe <= 0)) return MC_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (C % 40 == 0 && (C / 40 == 8 || C / 40 == 16 || C / 40 == 32)) {
        const int lpr = C / 40, rows_per_block = 4 * (64 / lpr);
        dim3 g5((M + rows_per_block - 1) / rows_per_block);
#define LN5(L) MC_LAUNCH(ln_fwd5_kernel<L>, g5, dim3(256), 0, s, (const half_t*)x, ldx, (half_t*)y, ldy, gamma, beta, pe, hw, \
                         nframes_pe, stats, M, C, eps)
        if (lpr == 8) LN5(8);
        else if (lpr == 16) LN5(16);
        else LN5(32);
#undef LN5
        return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
    }
    dim3 grid((M + 4 * LN_RPW - 1) / (4 * LN_RPW)), block(256);
    int nv = (C / 8 + 63) / 64;
    if (nv == 1)
        MC_LAUNCH(ln_fwd_kernel<1>, grid, block, 0, s, (const half_t*)x, ldx, (half_t*)y, ldy, gamma, beta, pe,
                  hw, nframes_pe, stats, M, C, eps);
    else if (nv == 2)
        MC_LAUNCH(ln_fwd_kernel<2>, grid, block, 0, s, (const half_t*)x, ldx, (half_t*)y, ldy, gamma, beta, pe,
                  hw, nframes_pe, stats, M, C, eps);
    else
        MC_LAUNCH(ln_fwd_kernel<3>, grid, block, 0, s, (const half_t*)x, ldx, (half_t*)y, ldy, gamma, beta, pe,
                  hw, nframes_pe, stats, M, C, eps);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

extern "C" int mc_layernorm_bwd_f16(const void* dy, int lddy, const void* x, int ldx, const float* stats,
                                    const float* gamma, const void* add, int ldadd, void* dx, int lddx,
                                    int M, int C, void* stream) {
    if (M <= 0 || C <= 0 || C % 8 || ldx % 8 || lddy % 8 || lddx % 8 || C > 1536) return MC_ERR_SHAPE;
    if (add && ldadd % 8) return MC_ERR_SHAPE;
    dim3 grid((M + 4 * LN_RPW - 1) / (4 * LN_RPW)), block(256);
    hipStream_t s = (hipStream_t)stream;
    int nv = (C / 8 + 63) / 64;
    if (nv == 1)
        MC_LAUNCH(ln_bwd_kernel<1>, grid, block, 0, s, (const half_t*)dy, lddy, (const half_t*)x, ldx, stats,
                  gamma, (const half_t*)add, ldadd, (half_t*)dx, lddx, M, C);
    else if (nv == 2)
        MC_LAUNCH(ln_bwd_kernel<2>, grid, block, 0, s, (const half_t*)dy, lddy, (const half_t*)x, ldx, stats,
                  gamma, (const half_t*)add, ldadd, (half_t*)dx, lddx, M, C);
    else
        MC_LAUNCH(ln_bwd_kernel<3>, grid, block, 0, s, (const half_t*)dy, lddy, (const half_t*)x, ldx, stats,
                  gamma, (const half_t*)add, ldadd, (half_t*)dx, lddx, M, C);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}
