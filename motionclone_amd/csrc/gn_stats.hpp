// GroupNorm statistics shared by norm.hip (apply / backward passes) and gemm4.hip (GroupNorm folded into the K = 320 GEMM).
#pragma once
#include "mc_common.hpp"

namespace mc {

// Round 3: the finalize pass INSIDE its consumers.  The 64 group totals of this block's frame are summed from the per-chunk
// partials by 64 threads (chunk order 0 .. nchunk-1, the order gn_finalize_kernel uses; both loops are kept from being reassociated) and turned into
// (mean, rstd) [mode 0] or (s1 / n, s2 / n) [mode 1] in LDS; chunk 0 of every frame also writes them to `stats_out` (the backward
// and the C ABI's stats tensor).  One launch less per GroupNorm (3.5 k launches per video).
__device__ __forceinline__ void gn_block_stats(const float* partial, int frame, int nchunk, float n, float eps, int mode,
                                               float* lds /* [64] */, float* stats_out, bool write) {
    const int t = threadIdx.x;
    if (t < 64) {
        // same order of additions as the finalize kernel, but eight loads in flight (the chunk loop is pure L2 latency)
        const float* q = partial + (size_t)frame * nchunk * 64 + t;
        float a = 0.f;
        int c = 0;
        for (; c + 8 <= nchunk; c += 8) {
#pragma clang fp reassociate(off)   // fast-math would turn the eight ordered adds into a tree: different last bits than finalize
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = q[(size_t)(c + u) * 64];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; c < nchunk; ++c) {
#pragma clang fp reassociate(off)   // the remainder chunks in order too (an unrolled remainder would be re-associated alike)
            a += q[(size_t)c * 64];
        }
        lds[t] = a;
    }
    __syncthreads();
    float r0 = 0.f, r1 = 0.f;
    if (t < 32) {
        const float a = lds[2 * t], b = lds[2 * t + 1];
        if (mode == 0) {
            const float mean = a / n;
            const float var = fmaxf(b / n - mean * mean, 0.f);
            r0 = mean;
            r1 = 1.0f / sqrtf(var + eps);
        } else {
            r0 = a / n;
            r1 = b / n;
        }
    }
    __syncthreads();
    if (t < 32) {
        lds[2 * t] = r0;
        lds[2 * t + 1] = r1;
        if (write) {
            stats_out[((size_t)frame * 32 + t) * 2] = r0;
            stats_out[((size_t)frame * 32 + t) * 2 + 1] = r1;
        }
    }
    __syncthreads();
}

}  // namespace mc
