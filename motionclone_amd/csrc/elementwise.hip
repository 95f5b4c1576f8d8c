// HBM-bound glue of the guided DDIM step (§8a A12):
// GEGLU (diffusers FeedForward used at attention.py:211 / motion_module.py:209),
// strided adds, 2x2 sum-pool (data-gradient of the nearest-2x upsample of
// resnet.py:65), latent <-> channels-last conversion, sinusoidal timestep
// embedding (unet.py:386-392), SiLU, and the fused classifier-free-guidance +
// guided DDIM update (motionclone_functions.py:239,255,326-389).
#include "mc_common.hpp"

namespace mc {

__global__ void geglu_fwd_kernel(const half_t* in, int ldi, half_t* out, int ldo, int M, int D) {
    const int VD = D / 8;
    const long total = (long)M * VD;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        long m = idx / VD;
        int c = (int)(idx - m * VD) * 8;
        half8_t h = ld8(in + m * ldi + c);
        half8_t g = ld8(in + m * ldi + D + c);
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = to_half((float)h[e] * gelu_f((float)g[e]));
        st8(out + m * ldo + c, o);
    }
}

__global__ void geglu_bwd_kernel(const half_t* dout, int lddo, const half_t* in, int ldi, half_t* din,
                                 int lddi, int M, int D) {
    const int VD = D / 8;
    const long total = (long)M * VD;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        long m = idx / VD;
        int c = (int)(idx - m * VD) * 8;
        half8_t h = ld8(in + m * ldi + c);
        half8_t g = ld8(in + m * ldi + D + c);
        half8_t d = ld8(dout + m * lddo + c);
        half8_t oh, og;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float gv = (float)g[e], dv = (float)d[e];
            oh[e] = to_half(dv * gelu_f(gv));
            og[e] = to_half(dv * (float)h[e] * gelu_grad_f(gv));
        }
        st8(din + m * lddi + c, oh);
        st8(din + m * lddi + D + c, og);
    }
}

// out = sa * a + sb * b over an [M][C] window of strided matrices (b may be null)
__global__ void add_kernel(const half_t* a, int lda, const half_t* b, int ldb, half_t* out, int ldo, int M,
                           int C, float sa, float sb) {
    const int VC = C / 8;
    const long total = (long)M * VC;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        long m = idx / VC;
        int c = (int)(idx - m * VC) * 8;
        half8_t x = ld8(a + m * lda + c);
        half8_t o;
        if (b) {
            half8_t y = ld8(b + m * ldb + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = to_half(sa * (float)x[e] + sb * (float)y[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = to_half(sa * (float)x[e]);
        }
        st8(out + m * ldo + c, o);
    }
}

// out[f, y, x, :] (+)= sum_{a,b in 0..1} in[f, 2y+a, 2x+b, :]
__global__ void sumpool2_kernel(const half_t* in, int ldi, half_t* out, int ldo, int frames, int H, int W,
                                int C, int accumulate) {
    const int VC = C / 8;
    const long total = (long)frames * H * W * VC;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        long tok = idx / VC;
        int c = (int)(idx - tok * VC) * 8;
        int x = (int)(tok % W);
        long t2 = tok / W;
        int y = (int)(t2 % H);
        long f = t2 / H;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        if (accumulate) {
            half8_t p = ld8(out + tok * ldo + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = (float)p[e];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                long src = (f * 2 * H + 2 * y + a) * 2 * W + 2 * x + b;
                half8_t v = ld8(in + src * ldi + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
            }
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = to_half(acc[e]);
        st8(out + tok * ldo + c, o);
    }
}

// [B, CL, F, H, W] fp16 -> [(b f h w), CP] channels-last, channels >= CL zero-filled
__global__ void latent_to_cl_kernel(const half_t* lat, half_t* out, int B, int CL, int F, int HW, int CP) {
    const long total = (long)B * F * HW * CP;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        int c = (int)(idx % CP);
        long tok = idx / CP;
        int p = (int)(tok % HW);
        long bf = tok / HW;
        int f = (int)(bf % F);
        int b = (int)(bf / F);
        half_t v = (half_t)0.f;
        if (c < CL) v = lat[(((size_t)b * CL + c) * F + f) * HW + p];
        out[idx] = v;
    }
}

// channels-last [(b f h w), ld] (first CL channels) -> [B, CL, F, H, W]; fp16 or fp32 output, scaled
__global__ void cl_to_latent_kernel(const half_t* in, int ld, void* out, int out_f32, float scale, int B, int CL,
                                    int F, int HW) {
    const long total = (long)B * CL * F * HW;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        int p = (int)(idx % HW);
        long r = idx / HW;
        int f = (int)(r % F);
        r /= F;
        int c = (int)(r % CL);
        int b = (int)(r / CL);
        float v = (float)in[(((size_t)b * F + f) * HW + p) * ld + c] * scale;
        if (out_f32)
            reinterpret_cast<float*>(out)[idx] = v;
        else
            reinterpret_cast<half_t*>(out)[idx] = to_half(v);
    }
}

// ---- VAE decode helpers (AutoencoderKL around the loop) ------------------------------------
// In-place row softmax of an fp16 score matrix in fp32 (AttentionBlock: softmax(scores.float()).type(dtype)).
// One 256-thread block per row, 16-byte vectors; three passes over a row that lives in L2.
__global__ __launch_bounds__(256) void softmax_rows_kernel(half_t* x, int ld, int cols) {
    __shared__ float red[8];
    half_t* row = x + (size_t)blockIdx.x * ld;
    const int nv = cols / 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float mx = -INFINITY;
    for (int v = threadIdx.x; v < nv; v += 256) {
        half8_t h = ld8(row + v * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)h[e]);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int v = threadIdx.x; v < nv; v += 256) {
        half8_t h = ld8(row + v * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += expf((float)h[e] - mx);
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    for (int v = threadIdx.x; v < nv; v += 256) {
        half8_t h = ld8(row + v * 8), o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)(expf((float)h[e] - mx) * inv);
        st8(row + v * 8, o);
    }
}

// decode_latents tail (pipeline_animation.py:260-262): channels-last frames -> [C, F, H, W] fp32, (x / 2 + 0.5).clamp(0, 1)
__global__ void video_post_kernel(const half_t* in, int ld, float* out, int C, int F, int HW) {
    const long total = (long)C * F * HW;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        int p = (int)(idx % HW);
        long r = idx / HW;
        int f = (int)(r % F);
        int c = (int)(r / F);
        float v = (float)in[((size_t)f * HW + p) * ld + c] * 0.5f + 0.5f;
        out[idx] = fminf(fmaxf(v, 0.f), 1.f);
    }
}

// DiagonalGaussianDistribution.sample / .mode (diffusers 0.16.0 models/vae.py) from the quant_conv moment tokens
// [(f p), 2*LAT] = (mean | logvar): out[f][c][p] = mean + exp(0.5 * clamp(logvar, -30, 20)) * noise[f][c][p]
__global__ void vae_sample_kernel(const half_t* mom, int ld, const half_t* noise, half_t* out, int n, int LAT, int HW) {
    const long total = (long)n * LAT * HW;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        int p = (int)(idx % HW);
        long r = idx / HW;
        int c = (int)(r % LAT);
        int f = (int)(r / LAT);
        const half_t* t = mom + ((size_t)f * HW + p) * ld;
        float v = (float)t[c];
        if (noise) {
            float lv = fminf(fmaxf((float)t[LAT + c], -30.f), 20.f);
            v += expf(0.5f * lv) * (float)noise[idx];
        }
        out[idx] = to_half(v);
    }
}

// Reference-video front end after the decoder (reference motionclone/utils/util.py:232-238): frames uint8 [N, Hs, Ws, 3]
// (decord layout) -> bilinear resize with align_corners=True -> / 127.5 - 1 -> fp16 [N, 3, H, W].  The reference
// interpolates the uint8 tensor, i.e. the resized value is re-quantised to 0..255 before normalisation; here it is
// rounded to nearest (torch's uint8 path is fixed-point and differs from that by at most one level).
__global__ void video_resize_kernel(const uint8_t* in, half_t* out, int N, int Hs, int Ws, int H, int W, int quantise) {
    const long total = (long)N * 3 * H * W;
    const float sy = H > 1 ? (float)(Hs - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(Ws - 1) / (float)(W - 1) : 0.f;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        int x = (int)(idx % W);
        long r = idx / W;
        int y = (int)(r % H);
        r /= H;
        int c = (int)(r % 3);
        int n = (int)(r / 3);
        float fy = sy * y, fx = sx * x;
        int y0 = (int)fy, x0 = (int)fx;
        int y1 = min(y0 + 1, Hs - 1), x1 = min(x0 + 1, Ws - 1);
        float wy = fy - y0, wx = fx - x0;
        const uint8_t* f = in + (size_t)n * Hs * Ws * 3;
        float v00 = f[((size_t)y0 * Ws + x0) * 3 + c], v01 = f[((size_t)y0 * Ws + x1) * 3 + c];
        float v10 = f[((size_t)y1 * Ws + x0) * 3 + c], v11 = f[((size_t)y1 * Ws + x1) * 3 + c];
        float v = (1.f - wy) * ((1.f - wx) * v00 + wx * v01) + wy * ((1.f - wx) * v10 + wx * v11);
        if (quantise == 1) v = rintf(v);
        if (quantise == 3) v = floorf(v);
        out[idx] = (half_t)(v / 127.5f - 1.0f);
    }
}

// quantise == 2: BIT-EXACT with torch's CPU uint8 bilinear path (F.interpolate on the channels-last uint8 frames, which is what
// util.py:232-238 calls; verified against torch 2.10 in tests/test_kernels.py::test_video_resize): separable, HORIZONTAL pass first
// with its result rounded to uint8, then the vertical pass; taps (i0, i0 + 1) with weights (1 - l, l), l = frac(i (in - 1) / (out - 1))
// in double, as 16-bit fixed point with `prec` fractional bits (the largest precision whose biggest weight stays below 2^15,
// found on the host: resize_precision), w = (int)(weight 2^prec + 0.5), accumulator preset to 2^(prec - 1), result >> prec,
// clamped to 0..255.  One thread per output element evaluates the horizontal pass for its two source rows itself: no
// intermediate image, no workspace.
__device__ __forceinline__ void resize_taps(int i, double scale, int insz, int prec, int& i0, int& i1, int& w0, int& w1) {
    const double real = scale * (double)i;
    i0 = min((int)real, insz - 1);
    i1 = min(i0 + 1, insz - 1);
    const double l = real - (double)i0;
    const double one = (double)(1 << prec);
    w0 = (int)((1.0 - l) * one + 0.5);
    w1 = (int)(l * one + 0.5);
}
__global__ void video_resize_exact_kernel(const uint8_t* in, half_t* out, int N, int Hs, int Ws, int H, int W, double sy,
                                          double sx, int py, int px) {
    const long total = (long)N * 3 * H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        int x = (int)(idx % W);
        long r = idx / W;
        int y = (int)(r % H);
        r /= H;
        int c = (int)(r % 3);
        int n = (int)(r / 3);
        int x0, x1, wx0, wx1, y0, y1, wy0, wy1;
        resize_taps(x, sx, Ws, px, x0, x1, wx0, wx1);
        resize_taps(y, sy, Hs, py, y0, y1, wy0, wy1);
        const uint8_t* f = in + (size_t)n * Hs * Ws * 3;
        const int ra = min(max(((1 << (px - 1)) + f[((size_t)y0 * Ws + x0) * 3 + c] * wx0 + f[((size_t)y0 * Ws + x1) * 3 + c] * wx1) >> px, 0), 255);
        const int rb = min(max(((1 << (px - 1)) + f[((size_t)y1 * Ws + x0) * 3 + c] * wx0 + f[((size_t)y1 * Ws + x1) * 3 + c] * wx1) >> px, 0), 255);
        const int v = min(max(((1 << (py - 1)) + ra * wy0 + rb * wy1) >> py, 0), 255);
        out[idx] = (half_t)((float)v / 127.5f - 1.0f);
    }
}

// ---- CLIP text encoder helpers (transformers CLIPTextModel, reference pipeline_animation.py:160-247) ---------------
// embeddings: out[(b s), :] = token_embedding[ids[b][s]] + position_embedding[s]
__global__ void clip_embed_kernel(const long* ids, const half_t* tok, const half_t* pos, half_t* out, int B, int S, int C,
                                  int vocab) {
    const int nv = C / 8;
    const long total = (long)B * S * nv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        int v = (int)(idx % nv);
        long r = idx / nv;
        int sidx = (int)(r % S);
        long id = ids[r];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        half8_t a = ld8(tok + (size_t)id * C + v * 8), b = ld8(pos + (size_t)sidx * C + v * 8), o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = to_half((float)a[e] + (float)b[e]);
        st8(out + (size_t)r * C + v * 8, o);
    }
}
// quick_gelu: x * sigmoid(1.702 x)
__global__ void quick_gelu_kernel(const half_t* in, half_t* out, long n) {
    for (long idx = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; idx < n;
         idx += (long)gridDim.x * blockDim.x * 8) {
        half8_t a = ld8(in + idx), o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = (float)a[e];
            o[e] = to_half(x / (1.0f + expf(-1.702f * x)));
        }
        st8(out + idx, o);
    }
}

// diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]
__global__ void timestep_embed_kernel(const float* t, half_t* out, int B, int dim) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * dim) return;
    int b = idx / dim, i = idx % dim;
    int half_dim = dim / 2;
    int j = i < half_dim ? i : i - half_dim;
    float freq = expf(-logf(10000.0f) * (float)j / (float)half_dim);
    float a = t[b] * freq;
    out[idx] = (half_t)(i < half_dim ? cosf(a) : sinf(a));
}

__global__ void silu_kernel(const half_t* in, half_t* out, long n) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x)
        out[idx] = to_half(silu_f((float)in[idx]));
}

struct DdimCoef {
    float cfg;            // classifier-free guidance scale s in eps_c + s*(eps_c - eps_u)
    float sqrt_a_t;       // sqrt(alpha_bar_t)
    float sqrt_1m_a_t;    // sqrt(1 - alpha_bar_t)
    float sqrt_a_prev;    // sqrt(alpha_bar_prev)
    float sqrt_1m_a_prev; // sqrt(1 - alpha_bar_prev)   (eta = 0)
    float score_coef;     // guidance_scale * sqrt(1 - alpha_bar_t) * (1 / grad_scale)
};

// eps_c/eps_u: channels-last [(f h w), ld] (first CL channels); x, score, out: [1, CL, F, H, W]
__global__ void cfg_ddim_kernel(const half_t* eps_c, const half_t* eps_u, int ld, const half_t* x,
                                const float* score, half_t* out, half_t* eps_out, DdimCoef k, int CL, int F,
                                int HW) {
    const long total = (long)CL * F * HW;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        int p = (int)(idx % HW);
        long r = idx / HW;
        int f = (int)(r % F);
        int c = (int)(r / F);
        size_t src = ((size_t)f * HW + p) * ld + c;
        float ec = (float)eps_c[src], eu = (float)eps_u[src];
        float eps = ec + k.cfg * (ec - eu);
        float xv = (float)x[idx];
        float x0 = (xv - k.sqrt_1m_a_t * eps) / k.sqrt_a_t;     // uses the un-guided eps (quirk A12)
        float e2 = eps;
        if (score) e2 -= k.score_coef * score[idx];
        out[idx] = to_half(k.sqrt_a_prev * x0 + k.sqrt_1m_a_prev * e2);
        if (eps_out) eps_out[idx] = to_half(eps);
    }
}

// schedule_customized_step in full (motionclone_functions.py:285-409): every branch is an affine map of
// (sample, model_output, score, variance_noise), same [B, C, F, H, W] layout for all operands
struct DdimGeneral {
    float x0_s, x0_m;       // pred_original_sample = x0_s * sample + x0_m * model_output        (:338-352, by prediction_type)
    float ep_s, ep_m;       // pred_epsilon         = ep_s * sample + ep_m * model_output
    float clip;             // > 0: clamp pred_original_sample to [-clip, clip]                   (:356-360)
    int rederive;           // use_clipped_model_output: eps = (sample - sqrt_a * x0) / sqrt_b    (:367-369)
    float sqrt_a, sqrt_b;
    float score_coef;       // guidance_scale * sqrt(1 - alpha_t) (0 without a score)             (:375-383)
    float c_x0, c_dir, c_noise;   // prev = c_x0 * x0 + c_dir * eps + c_noise * variance_noise    (:386-405)
};
__global__ void ddim_general_kernel(const half_t* sample, const half_t* mo, const float* score, const half_t* noise,
                                    half_t* prev, half_t* x0_out, half_t* eps_out, DdimGeneral k, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float sv = (float)sample[i], mv = (float)mo[i];
        float x0 = k.x0_s * sv + k.x0_m * mv;
        float ep = k.ep_s * sv + k.ep_m * mv;
        if (k.clip > 0.f) x0 = fminf(fmaxf(x0, -k.clip), k.clip);
        if (k.rederive) ep = (sv - k.sqrt_a * x0) / k.sqrt_b;
        if (eps_out) eps_out[i] = to_half(ep);                 // return_middle hands out the un-guided epsilon (:371-372)
        if (score) ep -= k.score_coef * score[i];
        float pv = k.c_x0 * x0 + k.c_dir * ep;
        if (noise) pv += k.c_noise * (float)noise[i];
        if (prev) prev[i] = to_half(pv);
        if (x0_out) x0_out[i] = to_half(x0);
    }
}

static inline int ew_blocks(long total) {
    long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace mc

using namespace mc;

extern "C" int mc_geglu_fwd_f16(const void* in, int ldi, void* out, int ldo, int M, int D, void* stream) {
    if (M <= 0 || D <= 0 || D % 8 || ldi % 8 || ldo % 8) return MC_ERR_SHAPE;
    MC_LAUNCH(geglu_fwd_kernel, dim3(ew_blocks((long)M * D / 8)), dim3(256), 0, (hipStream_t)stream,
              (const half_t*)in, ldi, (half_t*)out, ldo, M, D);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

extern "C" int mc_geglu_bwd_f16(const void* dout, int lddo, const void* in, int ldi, void* din, int lddi,
                                int M, int D, void* stream) {
    if (M <= 0 || D <= 0 || D % 8 || ldi % 8 || lddo % 8 || lddi % 8) return MC_ERR_SHAPE;
    MC_LAUNCH(geglu_bwd_kernel, dim3(ew_blocks((long)M * D / 8)), dim3(256), 0, (hipStream_t)stream,
              (const half_t*)dout, lddo, (const half_t*)in, ldi, (half_t*)din, lddi, M, D);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

extern "C" int mc_add_f16(const void* a, int lda, const void* b, int ldb, void* out, int ldo, int M, int C,
                          float sa, float sb, void* stream) {
    if (M <= 0 || C <= 0 || C % 8 || lda % 8 || ldo % 8 || (b && ldb % 8)) return MC_ERR_SHAPE;
    MC_LAUNCH(add_kernel, dim3(ew_blocks((long)M * C / 8)), dim3(256), 0, (hipStream_t)stream,
              (const half_t*)a, lda, (const half_t*)b, ldb, (half_t*)out, ldo, M, C, sa, sb);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

extern "C" int mc_sumpool2_f16(const void* in, int ldi, void* out, int ldo, int frames, int H, int W, int C,
                               int accumulate, void* stream) {
    if (frames <= 0 || H <= 0 || W <= 0 || C % 8 || ldi % 8 || ldo % 8) return MC_ERR_SHAPE;
    MC_LAUNCH(sumpool2_kernel, dim3(ew_blocks((long)frames * H * W * C / 8)), dim3(256), 0,
              (hipStream_t)stream, (const half_t*)in, ldi, (half_t*)out, ldo, frames, H, W, C, accumulate);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

extern "C" int mc_latent_to_cl_f16(const void* lat, void* out, int B, int CL, int F, int HW, int CP,
                                   void* stream) {
    if (B <= 0 || CL <= 0 || F <= 0 || HW <= 0 || CP < CL) return MC_ERR_SHAPE;
    MC_LAUNCH(latent_to_cl_kernel, dim3(ew_blocks((long)B * F * HW * CP)), dim3(256), 0, (hipStream_t)stream,
              (const half_t*)lat, (half_t*)out, B, CL, F, HW, CP);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

extern "C" int mc_cl_to_latent_f16(const void* in, int ld, void* out, int out_f32, float scale, int B, int CL,
                                   int F, int HW, void* stream) {
    if (B <= 0 || CL <= 0 || F <= 0 || HW <= 0 || ld < CL) return MC_ERR_SHAPE;
    MC_LAUNCH(cl_to_latent_kernel, dim3(ew_blocks((long)B * CL * F * HW)), dim3(256), 0, (hipStream_t)stream,
              (const half_t*)in, ld, out, out_f32, scale, B, CL, F, HW);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

extern "C" int mc_timestep_embed_f16(const float* t, void* out, int B, int dim, void* stream) {
    if (B <= 0 || dim <= 0 || dim % 2) return MC_ERR_SHAPE;
    MC_LAUNCH(timestep_embed_kernel, dim3((B * dim + 255) / 256), dim3(256), 0, (hipStream_t)stream, t,
              (half_t*)out, B, dim);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

extern "C" int mc_silu_f16(const void* in, void* out, long n, void* stream) {
    if (n <= 0) return MC_ERR_SHAPE;
    MC_LAUNCH(silu_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, (const half_t*)in,
              (half_t*)out, n);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

extern "C" int mc_cfg_ddim_step_f16(const void* eps_c, const void* eps_u, int ld, const void* x,
                                    const float* score, void* out, void* eps_out, float cfg, float sqrt_a_t,
                                    float sqrt_1m_a_t, float sqrt_a_prev, float sqrt_1m_a_prev,
                                    float score_coef, int CL, int F, int HW, void* stream) {
    if (CL <= 0 || F <= 0 || HW <= 0 || ld < CL) return MC_ERR_SHAPE;
    DdimCoef k{cfg, sqrt_a_t, sqrt_1m_a_t, sqrt_a_prev, sqrt_1m_a_prev, score_coef};
    MC_LAUNCH(cfg_ddim_kernel, dim3(ew_blocks((long)CL * F * HW)), dim3(256), 0, (hipStream_t)stream,
              (const half_t*)eps_c, (const half_t*)eps_u, ld, (const half_t*)x, score, (half_t*)out,
              (half_t*)eps_out, k, CL, F, HW);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

extern "C" int mc_ddim_step_general_f16(const void* sample, const void* model_output, const float* score,
                                        const void* noise, void* prev, void* x0_out, void* eps_out, long n,
                                        float x0_s, float x0_m, float ep_s, float ep_m, float clip, int rederive,
                                        float sqrt_a, float sqrt_b, float score_coef, float c_x0, float c_dir,
                                        float c_noise, void* stream) {
    if (n <= 0 || !sample || !model_output || (!prev && !x0_out && !eps_out)) return MC_ERR_SHAPE;
    if (rederive && sqrt_b <= 0.f) return MC_ERR_SHAPE;
    DdimGeneral k{x0_s, x0_m, ep_s, ep_m, clip, rederive, sqrt_a, sqrt_b, score_coef, c_x0, c_dir, c_noise};
    MC_LAUNCH(ddim_general_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, (const half_t*)sample,
              (const half_t*)model_output, score, (const half_t*)noise, (half_t*)prev, (half_t*)x0_out,
              (half_t*)eps_out, k, n);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

extern "C" int mc_version(void) { return 1; }

// ---- workspace sizes (bytes) of the entry points that take a caller-owned workspace -----------------------------------
extern "C" long mc_workspace_bytes_gemm_splitk(int M, int N, int splits) {
    // fp32 partial sums of `splits` K ranges over whole tiles (256 x 320: the slabs of gemm5 are tile-shaped)
    if (M <= 0 || N <= 0 || splits < 1) return -1;
    const long mp = ((long)M + 255) / 256 * 256, np = ((long)N + 319) / 320 * 320;
    return (long)sizeof(float) * splits * mp * np;
}
extern "C" long mc_workspace_bytes_attn_bwd(int nbatch, int heads, int Nq) {
    return (nbatch <= 0 || heads <= 0 || Nq <= 0) ? -1 : (long)sizeof(float) * nbatch * heads * Nq;
}
extern "C" long mc_workspace_bytes_tattn_loss(int B, int HW, int heads) {
    return (B <= 0 || HW <= 0 || heads <= 0) ? -1 : (long)sizeof(float) * B * HW * heads;
}

extern "C" int mc_softmax_rows_f16(void* x, int ld, int rows, int cols, void* stream) {
    if (rows <= 0 || cols <= 0 || cols % 8 || ld % 8 || ld < cols) return MC_ERR_SHAPE;
    MC_LAUNCH(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (half_t*)x, ld, cols);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

extern "C" int mc_video_post_f32(const void* in, int ld, float* out, int C, int F, int HW, void* stream) {
    if (C <= 0 || F <= 0 || HW <= 0 || ld < C) return MC_ERR_SHAPE;
    MC_LAUNCH(video_post_kernel, dim3(ew_blocks((long)C * F * HW)), dim3(256), 0, (hipStream_t)stream,
              (const half_t*)in, ld, out, C, F, HW);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

extern "C" int mc_vae_sample_f16(const void* moments, int ld, const void* noise, void* out, int n, int LAT, int HW,
                                 void* stream) {
    if (n <= 0 || LAT <= 0 || HW <= 0 || ld < 2 * LAT) return MC_ERR_SHAPE;
    MC_LAUNCH(vae_sample_kernel, dim3(ew_blocks((long)n * LAT * HW)), dim3(256), 0, (hipStream_t)stream,
              (const half_t*)moments, ld, (const half_t*)noise, (half_t*)out, n, LAT, HW);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

// weight precision of one axis of the exact resize (see video_resize_exact_kernel): plain double arithmetic on the host
static int resize_precision(int insz, int outsz, double scale) {
    double wt_max = 0.0;
    for (int i = 0; i < outsz; ++i) {
        const double real = scale * (double)i;
        const int i0 = std::min((int)real, insz - 1);
        const double l = real - (double)i0;
        wt_max = std::max(wt_max, std::max(l, 1.0 - l));
    }
    int prec = 0;
    for (prec = 0; prec < 22; ++prec)
        if ((int)(0.5 + wt_max * (double)(1 << (prec + 1))) >= (1 << 15)) break;
    return prec;
}

extern "C" int mc_video_resize_u8_f16(const void* in, void* out, int N, int Hs, int Ws, int H, int W, int quantise,
                                      void* stream) {
    if (N <= 0 || Hs <= 0 || Ws <= 0 || H <= 0 || W <= 0 || quantise < 0 || quantise > 3) return MC_ERR_SHAPE;
    if (quantise == 2) {
        const double sy = H > 1 ? (double)(Hs - 1) / (double)(H - 1) : 0.0, sx = W > 1 ? (double)(Ws - 1) / (double)(W - 1) : 0.0;
        MC_LAUNCH(video_resize_exact_kernel, dim3(ew_blocks((long)N * 3 * H * W)), dim3(256), 0, (hipStream_t)stream,
                  (const uint8_t*)in, (half_t*)out, N, Hs, Ws, H, W, sy, sx, resize_precision(Hs, H, sy), resize_precision(Ws, W, sx));
        return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
    }
    MC_LAUNCH(video_resize_kernel, dim3(ew_blocks((long)N * 3 * H * W)), dim3(256), 0, (hipStream_t)stream,
              (const uint8_t*)in, (half_t*)out, N, Hs, Ws, H, W, quantise);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

extern "C" int mc_clip_embed_f16(const long* ids, const void* tok, const void* pos, void* out, int B, int S, int C,
                                 int vocab, void* stream) {
    if (B <= 0 || S <= 0 || C <= 0 || C % 8 || vocab <= 0) return MC_ERR_SHAPE;
    MC_LAUNCH(clip_embed_kernel, dim3(ew_blocks((long)B * S * (C / 8))), dim3(256), 0, (hipStream_t)stream, ids,
              (const half_t*)tok, (const half_t*)pos, (half_t*)out, B, S, C, vocab);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

extern "C" int mc_quick_gelu_f16(const void* in, void* out, long n, void* stream) {
    if (n <= 0 || n % 8) return MC_ERR_SHAPE;
    MC_LAUNCH(quick_gelu_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0, (hipStream_t)stream, (const half_t*)in,
              (half_t*)out, n);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}
