/*
 * mc_kernels.h - C ABI of libmotionclone_hip.so (gfx950 / MI355X).
 *
 * The reference (LPengYang/MotionClone @ 2024-10-16) has no native layer: its hot
 * path reaches the GPU only through PyTorch / xformers operators.  This header is
 * the boundary a maintainer binds instead (ctypes stub: INTEGRATION.md); each entry
 * point names the reference operator(s) it replaces (paths relative to the
 * reference repository root).
 *
 * Conventions:
 *   - all activations fp16, channels-last token matrices [tokens, C] with an explicit
 *     row stride (ld*, in elements); token order is (batch, frame, y, x);
 *   - small parameter vectors (bias, gamma, beta, PE table) fp32;
 *   - the caller owns every buffer including workspaces; the library allocates
 *     nothing, never synchronises, launches only on `stream` (a hipStream_t), keeps
 *     no mutable global state, and is graph-capturable;
 *   - return value: 0 = MC_OK, -1 = bad shape/stride/alignment, -2 = unsupported
 *     size, -3 = launch failure.  Pointers must be 16-byte aligned.
 */
#ifndef MC_KERNELS_H
#define MC_KERNELS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MC_ABI_VERSION 1
int mc_version(void);

/* Workspace sizes in bytes (the caller owns every workspace; negative = bad arguments):
 *   gemm_splitk          `ws` of mc_gemm_splitk_f16 (fp32 partial tiles, splits from mc_gemm_splitk_plan & 0xFF)
 *   groupnorm            `partial` of mc_groupnorm_stats_f16 and mc_groupnorm_bwd_f16
 *   groupnorm_bwd_stats  `bstats` of mc_groupnorm_bwd_f16
 *   attn_bwd             `Dbuf` of mc_attn_bwd_f16
 *   tattn_loss           `unit_loss` of mc_tattn_loss_f16 */
long mc_workspace_bytes_gemm_splitk(int M, int N, int splits);
long mc_workspace_bytes_groupnorm(int frames, int hw);
long mc_workspace_bytes_groupnorm_bwd_stats(int frames);
long mc_workspace_bytes_attn_bwd(int nbatch, int heads, int Nq);
long mc_workspace_bytes_tattn_loss(int B, int HW, int heads);

/* ---- MFMA GEMM / implicit convolution ------------------------------------------------------
 * C[M,N] = alpha * A[M,K] . W[N,K]^T + bias[m / rows_per_batch][n] + R[m][n]
 * mode 0 DENSE    : nn.Linear / 1x1 conv  (attention.py:65,93,355-357,364; motion_module.py:113,135;
 *                   resnet.py:181; diffusers FeedForward) and their data-gradients (W pre-transposed)
 * mode 1 CONV_S1  : 3x3 stride 1 pad 1     (resnet.py:148,168; unet.py:98,249) and its data-gradient
 * mode 2 CONV_S2  : 3x3 stride 2 pad 1     (resnet.py:94)
 * mode 3 CONV_UP  : F.interpolate(nearest, 2x) + 3x3 (resnet.py:65,78), upsample never materialised
 * mode 4 TCONV_S2 : data-gradient of mode 2
 * A2 (optional) supplies channels [c1, ctot) - the skip concat of unet_blocks.py:634,740.
 * K = ctot (dense) or 9*ctot (conv; k = (c/64)*576 + tap*64 + c%64: channel-tile major, tap minor).
 * K, ctot, c1 multiples of 64; N, ldc multiples of 4.
 * flags: bits 0-7 block tile edge (0 = auto, 64, 128); 0x100 = (removed) first-generation kernel -> unsupported;
 *        0x200 = fused GEGLU epilogue: W rows interleaved (h_j, gate_j), C gets N/2 columns h_j * gelu(gate_j);
 *        0x400 = 3-stage staging ring (128/64 tiles); 0x800 = mode 2 with padding only right / bottom
 *        (diffusers Downsample2D(padding=0): F.pad(x, (0,1,0,1)) + stride-2 conv, the VAE encoder);
 *        bits 12-15 = large-tile geometry of gemm3.hip (0 = automatic; 10 = the K = 320 streaming kernel gemm4.hip);
 *        bits 16-19 = gemm4: workgroups per 256-row block (0 = automatic);
 *        bits 20-21 = share: 2^share independent launch sequences are in flight on other streams, so the automatic
 *        tile / split-K choice targets 256 >> share CUs (mc_gemm_splitk_plan takes the same value in mode bits 8-9). */
int mc_gemm_f16(const void* A, const void* A2, const void* W, void* C, const void* R, const float* bias,
                int M, int N, int K, int lda, int lda2, int ldc, int ldr, int c1, int ctot, int mode,
                int Hs, int Ws, int Ho, int Wo, int rows_per_batch, float alpha, int flags, void* stream);

/* mc_gemm_f16 that ALSO leaves the GroupNorm(32) partial sums of its OUTPUT (round 6): (sum, sum of squares) of the rounded fp16
 * values per (frame, chunk of 64 or 32 rows, group), written by the epilogue of the one-pass 256x320 / 128x320 ring kernels, so
 * that the GroupNorm reading C next needs no statistics pass (one launch and one read of the tensor less per GroupNorm:
 * resnet.py:197 after conv1; resnet.py:186 / attention.py:105 / motion_module.py:145 after conv2 + shortcut).
 *   gn_partial: float[(M / gn_hw) * (gn_hw / 32) * 64] = mc_workspace_bytes_gemm_gnstats(M / gn_hw, gn_hw); gn_hw: tokens per frame.
 * Returns the chunk height used (64 or 32: hand gn_hw / it to mc_groupnorm_fwd_partial_f16 as `pchunks`), or MC_ERR_UNSUPPORTED
 * (-2) with NOTHING launched - the library's own choice for the problem is another kernel (split-K, K = 320 streaming, small
 * tiles), N / 32 is not 10 / 20 / 40, or a wave tile would straddle frames: call mc_gemm_f16 + mc_groupnorm_fwd_f16. */
long mc_workspace_bytes_gemm_gnstats(int frames, int hw);
int mc_gemm_gnstats_f16(const void* A, const void* A2, const void* W, void* C, const void* R, const float* bias,
                        int M, int N, int K, int lda, int lda2, int ldc, int ldr, int c1, int ctot, int mode,
                        int Hs, int Ws, int Ho, int Wo, int rows_per_batch, float alpha, int flags, float* gn_partial, int gn_hw,
                        void* stream);

/* Persistent tile loop over the 256x320 tile (gemm6.hip, round 6): the same product as mc_gemm_f16 (DENSE, bit-identical to
 * its 256x320 kernel) with ONE workgroup per CU walking the output tiles and the LDS operand ring running through tile
 * boundaries - for the Linear layers (diffusers FeedForward at attention.py:211,288 / motion_module.py:209,222; to_q|k|v and
 * to_out at attention.py:355-364; proj_in / proj_out at attention.py:65,93 and motion_module.py:113,135), where a tile's fixed
 * cost was up to a third of its k-loop.
 *   workspace: mc_workspace_bytes_gemm_tileloop(0) bytes, ZERO at kernel start: per-XCD tile counters of the dynamic tile order
 *     and the hand-over flags of stream-K; the kernel zeroes them again before it ends (no memset per launch; launches that may
 *     overlap need separate blocks); null = static tile order.
 *   partials: mc_workspace_bytes_gemm_tileloop(1) bytes of scratch, flags 0x2 (stream-K) only: the k-stages of each XCD's tile
 *     list are dealt evenly to its workgroups, tiles are cut along k where a range ends, the later pieces' fp32 sums meet the
 *     first piece's here (replaces split-K + reduce; a cut tile's sum is deterministic but not the one-chain sum).
 *   flags: 0x200 fused GEGLU; 0x2 stream-K; 0x1 A/B: drain the epilogue's stores before the next tile's first wait;
 *     0x4: tile order with the column blocks outermost (an XCD keeps sn weight panels in its L2 while all its row groups pass;
 *     default: keeps a row group's activation panels and streams the weights past them; same results); bits 16-23 grid cap / 8.
 * Returns MC_ERR_UNSUPPORTED (-2) outside its shapes (K < 256, N % 8, two-source A, per-batch bias, M < 1793 without
 * stream-K): call mc_gemm_f16. */
long mc_workspace_bytes_gemm_tileloop(int which);
int mc_gemm_tileloop_f16(const void* A, const void* A2, const void* W, void* C, const void* R, const float* bias,
                         int M, int N, int K, int lda, int lda2, int ldc, int ldr, int c1, int rows_per_batch, float alpha,
                         int flags, void* workspace, size_t ws_bytes, void* partials, size_t partial_bytes, void* stream);

/* PROFILING ONLY: the kernel structure used by the calling thread's last mc_gemm_f16 / mc_gemm_splitk_f16 call:
 * 2 / 20 = gemm2 128x128 / 64x64 tiles, 31..35 = gemm3 geometry 1..5, 4 = gemm4, 51 / 54 = gemm5 with 256- / 128-row tiles;
 * + 100 for a split-K call (bench.py names its roofline rows with it). */
int mc_gemm_last_kernel(void);
/* same for the spatial attention entries: 0 = register-staged kernels, 1 = LDS-DMA ring kernels (mc_attn_bwd_f16: bit 0 dQ, bit 1
 * dK/dV) - profiling / tests only (attention.hip) */
int mc_attn_last_kernel(void);
/* temporal attention: 1 = the calling thread's last mc_tattn_fwd_f16 / mc_tattn_bwd_f16 ran the 16-byte-load kernel (temporal.hip) */
int mc_tattn_last_kernel(void);
#ifdef MC_TOOLS
/* TOOLS BUILD ONLY (-DMC_TOOLS: tools/_build/libmotionclone_hip_tools.so and the host simulator of tests/): process-wide
 * debug state and environment A/B switches.  The product library exports none of these and reads no environment variable. */
/* timing experiments that drop parts of the GEMM kernels (1 = no global stores, 2 = no MFMA loop, 4 = no epilogue,
 * 8 = no operand staging); outputs are garbage while set.  0 restores normal operation. */
int mc_gemm_debug(int bits);
int mc_tattn_debug_buffer(void* device_buffer); /* intermediates of mc_tattn_bwd_f16 (F <= 16, d = 40), units*64*24 floats */
int mc_gemm_debug_buffer(void* device_buffer);   /* bit 16: in-kernel cycle stamps of gemm4 land here */
#endif

/* Norm + Linear in ONE launch for the K = 320 level (round 4): C[M,N] = norm(A[M,320]) W[N,320]^T + bias, the normalisation
 * applied to the rows of A in registers inside the streaming kernel (gemm4.hip) - the normalised tensor never exists in HBM.
 *   kind 1: LayerNorm(gamma, beta, eps) over the 320 channels, + pe[(row / hw) % nframes_pe][:] when pe != NULL
 *           (= mc_layernorm_fwd_f16 then mc_gemm_f16; reference attention.py:189,206,212 -> 355-364, motion_module.py:204-213);
 *           stats: float[M][2] (mean, rstd) for mc_layernorm_bwd_f16, or NULL.  With pe: hw % 256 == 0.
 *   kind 2: GroupNorm(32 groups, NO activation) over frames of hw tokens (= mc_groupnorm_fwd_f16(silu = 0) then mc_gemm_f16;
 *           Transformer3DModel.norm + proj_in, attention.py:61-65,105-117; motion_module.py:112-113,145-151); hw % 256 == 0;
 *           partial: workspace of mc_workspace_bytes_groupnorm(M / hw, hw); stats: float[(M / hw) * 32 * 2] out (for the backward).
 * flags: 0x200 = fused GEGLU epilogue (as mc_gemm_f16); bits 16-23 (kind 2) = n > 0: `partial` already holds the sums of A, written
 * with n chunks per frame by the kernel that produced A (mc_gemm_gnstats_f16) - no statistics pass.  Returns MC_ERR_UNSUPPORTED
 * (-2) for shapes outside that kernel: the caller then issues the two-launch form. */
int mc_norm_gemm_f16(const void* A, const void* W, void* C, const float* bias, int M, int N, int K, int lda, int ldc,
                     int kind, const float* gamma, const float* beta, const float* pe, int hw, int nframes_pe, float eps,
                     float* stats, float* partial, int flags, void* stream);

/* Split-K variant for small-M / deep-K problems (the 8x8 and 16x16-level 3x3 convs of unet_blocks.py:Downsample3D /
 * ResnetBlock3D at reference motionclone/models/resnet.py:110-209): K is cut into `splits` ranges computed by
 * separate workgroups into the fp32 workspace ws[splits][M][N]; a reduce kernel applies bias / residual.
 * mc_gemm_splitk_plan returns 1 (= call mc_gemm_f16 instead) or ranges | (geometry << 8): the number of K ranges and
 * the gemm3 geometry to request in flags bits 12-15 (1: 256x320 tiles, 4: 128x320 tiles). */
int mc_gemm_splitk_plan(int M, int N, int K, int mode);
int mc_gemm_splitk_f16(const void* A, const void* A2, const void* W, void* C, const void* R, const float* bias,
                       int M, int N, int K, int lda, int lda2, int ldc, int ldr, int c1, int ctot, int mode, int Hs,
                       int Ws, int Ho, int Wo, int rows_per_batch, float alpha, int flags, float* ws, int splits,
                       void* stream);

/* ---- GroupNorm(32) [+SiLU] ------------------------------------------------------------------
 * resnet.py:21-29,186-187,197-203; attention.py:61,105; motion_module.py:112,145; unet.py:245.
 * Two-source input (a: channels [0,c1), b: [c1,ctot)).  partial: float[frames*mc_gn_nchunk(hw)*64]
 * workspace; stats: float[frames*32*2] = (mean, rstd). */
int mc_gn_nchunk(int hw);
int mc_groupnorm_stats_f16(const void* a, const void* b, int lda, int ldb, int c1, int ctot, int frames, int hw,
                           float eps, float* partial, float* stats, void* stream);
int mc_groupnorm_apply_f16(const void* a, const void* b, int lda, int ldb, int c1, int ctot, int frames, int hw,
                           const float* stats, const float* gamma, const float* beta, void* out, int ldo,
                           int silu, void* stream);
/* statistics + normalisation (+ SiLU) of one GroupNorm in two launches (partial sums; apply, which finalises the statistics in
 * its prologue and writes them to `stats` for the backward): the results of mc_groupnorm_stats_f16 followed by
 * mc_groupnorm_apply_f16, without the finalize launch between them. */
int mc_groupnorm_fwd_f16(const void* a, const void* b, int lda, int ldb, int c1, int ctot, int frames, int hw, float eps,
                         float* partial, float* stats, const float* gamma, const float* beta, void* out, int ldo, int silu,
                         void* stream);
/* mc_groupnorm_fwd_f16 WITHOUT its statistics pass (round 6; single source, ONE launch): `partial` holds the per-chunk (sum, sum of
 * squares) of every (frame, group) as left by the epilogue of the kernel that produced `a` - mc_gemm_gnstats_f16 - with
 * `pchunks` = hw / (the chunk height that call returned) chunks per frame.  resnet.py:197-199 (norm2 after conv1),
 * resnet.py:186, attention.py:105 and motion_module.py:145 after a block's conv2 + shortcut. */
int mc_groupnorm_fwd_partial_f16(const void* a, int lda, int ctot, int frames, int hw, float eps, const float* partial,
                                 int pchunks, float* stats, const float* gamma, const float* beta, void* out, int ldo,
                                 int silu, void* stream);
/* data-gradient (autograd of the above; reference motionclone_functions.py:236). bstats: float[frames*64] */
int mc_groupnorm_bwd_f16(const void* a, const void* b, int lda, int ldb, int c1, int ctot, int frames, int hw,
                         const void* dz, int lddz, const float* stats, const float* gamma, const float* beta,
                         int silu, float* partial, float* bstats, void* dx, int lddx, int accumulate,
                         void* stream);

/* ---- LayerNorm ------------------------------------------------------------------------------
 * attention.py:189,206,212; motion_module.py:204,210.  pe (optional, float[nframes_pe][C]) is the
 * sinusoidal temporal table of motion_module.py:237-246, added to row m at frame (m / hw) % nframes_pe
 * (motion_module.py:279-282).  stats: float[M*2] = (mean, rstd), may be NULL. */
int mc_layernorm_fwd_f16(const void* x, int ldx, void* y, int ldy, const float* gamma, const float* beta,
                         const float* pe, int hw, int nframes_pe, float* stats, int M, int C, float eps,
                         void* stream);
int mc_layernorm_bwd_f16(const void* dy, int lddy, const void* x, int ldx, const float* stats,
                         const float* gamma, const void* add, int ldadd, void* dx, int lddx, int M, int C,
                         void* stream);

/* ---- spatial self / text cross attention (flash-style) --------------------------------------
 * attention.py:387-490 and the xformers slot :535-542.  Row of (batch b, i): q: b*Nq+i, k/v:
 * (b / kv_bdiv)*Nk + j; head h occupies columns [h*d, (h+1)*d).  lse: float[nbatch*heads*Nq]. */
int mc_attn_fwd_f16(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, void* o, int ldo,
                    float* lse, int Nq, int Nk, int heads, int d, int nbatch, int kv_bdiv, float scale,
                    void* stream);
/* causal self-attention forward, N queries = N keys (the CLIP text encoder of pipeline_animation.py:160-247,
 * transformers CLIPTextModel: key j is masked for query i when j > i); no log-sum-exp output */
int mc_attn_fwd_causal_f16(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, void* o, int ldo,
                           int N, int heads, int d, int nbatch, float scale, void* stream);
/* dq always; dk/dv when non-NULL (requires kv_bdiv == 1).  Dbuf: float[nbatch*heads*Nq] workspace. */
int mc_attn_bwd_f16(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, const void* o,
                    int ldo, const void* dO, int lddo, const float* lse, float* Dbuf, void* dq, int lddq,
                    void* dk, int lddk, void* dv, int lddv, int Nq, int Nk, int heads, int d, int nbatch,
                    int kv_bdiv, float scale, void* stream);

/* ---- temporal attention + MotionClone guidance ----------------------------------------------
 * motion_module.py:274-345 (VersatileAttention over the F frames of one spatial position);
 * unit (b, p, head) reads rows (b*F + f)*HW + p at column offset head*d.  F <= 32. */
int mc_tattn_fwd_f16(const void* q, const void* k, const void* v, int ld, void* o, int ldo, int B, int F,
                     int HW, int heads, int d, float scale, void* stream);
/* motionclone_functions.py:260-283 + torch.topk(k=1) of :79 -> top_val fp16 / top_idx u8, [B*HW, heads, F] */
int mc_tattn_top1_f16(const void* q, const void* k, int ld, void* top_val, void* top_idx, int B, int F, int HW,
                      int heads, int d, float scale, void* stream);
/* get_temp_attn_prob (motionclone_functions.py:260-283): prob fp16 [B*HW, heads, F, F] */
int mc_tattn_prob_f16(const void* q, const void* k, int ld, void* prob, int B, int F, int HW, int heads, int d,
                      float scale, void* stream);
/* motionclone_functions.py:85-100 for one module: loss[0] = mean((gather(P, idx) - ref)^2).
 * unit_loss: float[B*HW*heads] workspace. */
int mc_tattn_loss_f16(const void* q, const void* k, int ld, const void* ref_idx, const float* ref_val,
                      float* unit_loss, float* loss, int B, int F, int HW, int heads, int d, float scale,
                      void* stream);
/* data-gradient of the attention (dO may be NULL) fused with the guidance seed
 * dP[q, idx[q]] += seed_coef * (P[q, idx[q]] - ref[q]) (ref_idx may be NULL): motionclone_functions.py:236 */
int mc_tattn_bwd_f16(const void* q, const void* k, const void* v, int ld, const void* dO, int lddo, void* dq,
                     void* dk, void* dv, int ldg, const void* ref_idx, const float* ref_val, float seed_coef,
                     int B, int F, int HW, int heads, int d, float scale, void* stream);
int mc_reduce_sum_f32(const float* in, long n, float scale, float* out, void* stream);

/* ---- element-wise glue ------------------------------------------------------------------------ */
/* diffusers GEGLU: out = in[:, :D] * gelu(in[:, D:])  (attention.py:211, motion_module.py:209) */
int mc_geglu_fwd_f16(const void* in, int ldi, void* out, int ldo, int M, int D, void* stream);
int mc_geglu_bwd_f16(const void* dout, int lddo, const void* in, int ldi, void* din, int lddi, int M, int D,
                     void* stream);
/* out = sa*a + sb*b (b may be NULL) - residual / skip-gradient accumulation */
int mc_add_f16(const void* a, int lda, const void* b, int ldb, void* out, int ldo, int M, int C, float sa,
               float sb, void* stream);
/* data-gradient of F.interpolate(nearest, 2x) (resnet.py:65): 2x2 sum-pool */
int mc_sumpool2_f16(const void* in, int ldi, void* out, int ldo, int frames, int H, int W, int C,
                    int accumulate, void* stream);
/* [B, CL, F, H, W] latent <-> channels-last tokens (the only layout change on the path) */
int mc_latent_to_cl_f16(const void* lat, void* out, int B, int CL, int F, int HW, int CP, void* stream);
int mc_cl_to_latent_f16(const void* in, int ld, void* out, int out_f32, float scale, int B, int CL, int F,
                        int HW, void* stream);
/* diffusers Timesteps(dim, flip_sin_to_cos=True, freq_shift=0) (unet.py:101,386-391) */
int mc_timestep_embed_f16(const float* t, void* out, int B, int dim, void* stream);
int mc_silu_f16(const void* in, void* out, long n, void* stream);
/* VAE decode around the loop (AnimationPipeline.decode_latents, reference pipeline_animation.py:249-263; the VAE is
 * diffusers==0.16.0 AutoencoderKL): in-place fp32 row softmax of the single-head AttentionBlock's fp16 score matrix,
 * and the (x / 2 + 0.5).clamp(0, 1) float32 [C, F, H, W] video tail (:260-262) */
int mc_softmax_rows_f16(void* x, int ld, int rows, int cols, void* stream);
int mc_video_post_f32(const void* in, int ld, float* out, int C, int F, int HW, void* stream);
/* CLIP text encoder pieces: token + position embedding lookup, quick_gelu (x * sigmoid(1.702 x)) */
int mc_clip_embed_f16(const long* ids, const void* tok, const void* pos, void* out, int B, int S, int C, int vocab,
                      void* stream);
int mc_quick_gelu_f16(const void* in, void* out, long n, void* stream);
/* reference-video front end behind the decoder (util.py:232-238): uint8 frames [N, Hs, Ws, 3] -> bilinear
 * (align_corners=True) -> [N, 3, H, W] fp16 in [-1, 1].  quantise: 2 = bit-exact with torch's CPU uint8 bilinear resize (separable,
 * horizontal pass first rounded to uint8, 16-bit fixed-point weights: what the reference's F.interpolate on the uint8 frames computes),
 * 0 = float result, 1 = float result rounded to the nearest level, 3 = float result truncated */
int mc_video_resize_u8_f16(const void* in, void* out, int N, int Hs, int Ws, int H, int W, int quantise, void* stream);
/* latent_dist.sample() / .mode() of AutoencoderKL.encode (motionclone_functions.py:64,125): moment tokens
 * [(f p), 2*LAT] (mean | logvar) and an optional standard-normal draw [n, LAT, HW] -> [n, LAT, HW] */
int mc_vae_sample_f16(const void* moments, int ld, const void* noise, void* out, int n, int LAT, int HW, void* stream);
/* eps = eps_c + cfg*(eps_c - eps_u) (motionclone_functions.py:239,255) followed by the guided DDIM
 * update of schedule_customized_step (:326-389, eta = 0):
 *   x0 = (x - sqrt(1-a_t) eps) / sqrt(a_t);  eps' = eps - score_coef * score;
 *   out = sqrt(a_prev) x0 + sqrt(1-a_prev) eps'.
 * eps_c / eps_u channels-last [(f h w), ld]; x, score (fp32, may be NULL), out: [1, CL, F, H, W]. */
int mc_cfg_ddim_step_f16(const void* eps_c, const void* eps_u, int ld, const void* x, const float* score,
                         void* out, void* eps_out, float cfg, float sqrt_a_t, float sqrt_1m_a_t,
                         float sqrt_a_prev, float sqrt_1m_a_prev, float score_coef, int CL, int F, int HW,
                         void* stream);
/* schedule_customized_step with every branch (motionclone_functions.py:285-409): prediction_type epsilon / sample /
 * v_prediction, clip_sample, use_clipped_model_output, eta > 0 with variance noise, the score term, return_middle.
 *   x0  = x0_s * sample + x0_m * model_output, clamped to [-clip, clip] if clip > 0;
 *   eps = ep_s * sample + ep_m * model_output, or (sample - sqrt_a x0) / sqrt_b if rederive;   -> eps_out (un-guided)
 *   eps' = eps - score_coef * score;   prev = c_x0 x0 + c_dir eps' + c_noise noise.
 * All tensors n contiguous elements in one common layout; fp16 except score (fp32); score / noise / each output may be NULL. */
int mc_ddim_step_general_f16(const void* sample, const void* model_output, const float* score, const void* noise,
                             void* prev, void* x0_out, void* eps_out, long n, float x0_s, float x0_m, float ep_s,
                             float ep_m, float clip, int rederive, float sqrt_a, float sqrt_b, float score_coef,
                             float c_x0, float c_dir, float c_noise, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MC_KERNELS_H */
