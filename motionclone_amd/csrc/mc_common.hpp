// Common device-side vocabulary for the MotionClone gfx950 kernels.
//
// Every kernel in this directory is written against the small set of wave-level
// primitives declared here (64-lane shuffles, two MFMA shapes, vector types).
// On the product build (hipcc --offload-arch=gfx950) they lower to the CDNA4
// builtins.  When MC_EMU is defined (tests/hipemu only, never shipped) the same
// sources are compiled for the host against a lane-accurate fiber simulator so
// that index math / barrier placement can be exercised without a GPU.
#pragma once

#ifdef MC_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>
#include <algorithm>
#include <type_traits>
#include <cmath>
#include <cstdio>
#include <cstdlib>

// The TEST / TOOLS builds (-DMC_TOOLS: tests/hipemu and tools/_build/libmotionclone_hip_tools.so, never loaded by the
// package) read A/B switches from the environment and export the profiling hooks (mc_gemm_debug* / mc_tattn_debug_buffer).
// The PRODUCT library has neither: no environment read, no mutable global - MC_ENV_INT folds to its default and the
// variable name does not even reach the binary (tests/test_abi.py checks `nm -D` and the string table).
#if defined(MC_EMU) && !defined(MC_TOOLS)
#define MC_TOOLS 1
#endif
#ifdef MC_TOOLS
static inline int mc_env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
#define MC_ENV_INT(name, dflt) mc_env_int(name, dflt)
#else
#define MC_ENV_INT(name, dflt) (dflt)
#endif

namespace mc {

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;

#ifndef MC_EMU
static thread_local int g_launch_error = 0;
static inline void note_launch(const char* what) {
    hipError_t e = hipGetLastError();
    g_launch_error = (int)e;
    if (e != hipSuccess) fprintf(stderr, "motionclone_hip: launch of %s failed: %s (%d)\n", what, hipGetErrorString(e), (int)e);
}
#endif

#ifdef MC_EMU
// ---- simulator lowering -------------------------------------------------------
__device__ inline float shfl_xor(float v, int m) { return hipemu::shfl_xor(v, m); }
__device__ inline int shfl_xor(int v, int m) { return hipemu::shfl_xor(v, m); }
__device__ inline float shfl(float v, int src) { return hipemu::shfl(v, src); }
__device__ inline int shfl(int v, int src) { return hipemu::shfl(v, src); }
__device__ inline f32x4 mfma16(half4_t a, half4_t b, f32x4 c) { return hipemu::mfma_16x16x16(a, b, c); }
__device__ inline f32x16 mfma32(half8_t a, half8_t b, f32x16 c) { return hipemu::mfma_32x32x16(a, b, c); }
__device__ inline f32x4 mfma16k32(half8_t a, half8_t b, f32x4 c) { return hipemu::mfma_16x16x32(a, b, c); }
__device__ inline void mfma32_agpr(f32x16& c, half8_t a, half8_t b) { c = hipemu::mfma_32x32x16(a, b, c); }
__device__ inline void mfma32_vgpr(f32x16& c, half8_t a, half8_t b) { c = hipemu::mfma_32x32x16(a, b, c); }
__device__ inline float fast_exp(float x) { return expf(x); }
__device__ inline float fast_rsqrt(float x) { return 1.0f / sqrtf(x); }
#define MC_DYN_SMEM(name) char* name = hipemu::dyn_smem()
#define MC_LAUNCH(kern, grid, block, smem, stream, ...) \
    hipemu::launch((grid), (block), (smem), [=]() { kern(__VA_ARGS__); })
#define MC_LAST_ERROR() 0
template <class K>
static inline void allow_big_smem(K, size_t) {}
#else
// ---- gfx950 lowering ----------------------------------------------------------
__device__ __forceinline__ float shfl_xor(float v, int m) { return __shfl_xor(v, m, 64); }
__device__ __forceinline__ int shfl_xor(int v, int m) { return __shfl_xor(v, m, 64); }
__device__ __forceinline__ float shfl(float v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ int shfl(int v, int src) { return __shfl(v, src, 64); }
// D[16x16] += A[16x16] * B[16x16]; lane l: a = A[l&15][4*(l>>4)+j], b = B[4*(l>>4)+j][l&15],
// c[i] = C[4*(l>>4)+i][l&15].
__device__ __forceinline__ f32x4 mfma16(half4_t a, half4_t b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
}
// D[32x32] += A[32x16] * B[16x32]; lane l: a = A[l&31][8*(l>>5)+j], b = B[8*(l>>5)+j][l&31],
// c[r] = C[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
__device__ __forceinline__ f32x16 mfma32(half8_t a, half8_t b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// D[16x16] += A[16x32] * B[32x16]; lane l: a = A[l&15][8*(l>>4)+j], b = B[8*(l>>4)+j][l&15], c as mfma16.
__device__ __forceinline__ f32x4 mfma16k32(half8_t a, half8_t b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// The same MFMA with the accumulator's register class chosen by the CALLER (inline asm: "a" = AGPR, "v" = VGPR).  A wave that owns
// a whole SIMD may use 256 + 256 registers, but hipcc puts every accumulator of a kernel into ONE class: 320 accumulators
// (128 x 160 wave tile) spilled 468 registers.  With four of five column blocks in AGPRs and one in VGPRs they fit.  No
// instruction reads an accumulator closer than a whole k-slice (20 MFMAs) behind its write: no MFMA hazard to pad.
__device__ __forceinline__ void mfma32_agpr(f32x16& c, half8_t a, half8_t b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma32_vgpr(f32x16& c, half8_t a, half8_t b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ float fast_rsqrt(float x) { return rsqrtf(x); }
#define MC_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
// hipGetLastError() is sticky per thread and PyTorch leaves benign errors behind: drop anything stale before a
// launch, and report the launch's own status with its name.
#define MC_LAUNCH(kern, grid, block, smem, stream, ...)                                   \
    do {                                                                                  \
        (void)hipGetLastError();                                                          \
        hipLaunchKernelGGL(kern, (grid), (block), (smem), (stream), __VA_ARGS__);         \
        mc::note_launch(#kern);                                                           \
    } while (0)
#define MC_LAST_ERROR() (mc::g_launch_error)
// dynamic LDS above 64 KiB has to be opted into per kernel (gfx950 has 160 KiB per CU)
template <class K>
static inline void allow_big_smem(K kern, size_t bytes) {
    if (bytes > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)bytes);
}
#endif

// ---- direct global -> LDS loads (LDS-DMA) ---------------------------------------------------------
// glds16(buf, voff, lds): every lane moves 16 bytes from byte offset `voff` of the buffer straight into LDS at
// (wave-uniform) `lds` + lane*16, without passing through VGPRs (buffer_load_dwordx4 ... lds).  An offset at or
// beyond the buffer's size reads as zeros - that is how conv padding and M/N tails are filled, in hardware.
// The data is visible to other waves after the issuing wave's vmcnt wait plus a workgroup barrier
// (__syncthreads() provides both).
#ifdef MC_EMU
struct GBuf {
    const char* base;
    uint32_t bytes;
};
__device__ inline GBuf make_gbuf(const void* p, uint32_t bytes) { return GBuf{(const char*)p, bytes}; }
__device__ inline void glds16(GBuf b, uint32_t voff, char* lds_wave_base) {
    char* d = lds_wave_base + hipemu::lane_id() * 16;
    if ((uint64_t)voff + 16 <= b.bytes)
        memcpy(d, b.base + voff, 16);
    else
        memset(d, 0, 16);
}
// 16-byte buffer load into registers with the same hardware range check (out of range -> zeros, no VALU select)
__device__ inline half8_t gbuf_ld8(GBuf b, uint32_t voff) {
    half8_t r;
    if ((uint64_t)voff + 16 <= b.bytes)
        memcpy(&r, b.base + voff, 16);
    else
        memset(&r, 0, 16);
    return r;
}
#else
typedef __amdgpu_buffer_rsrc_t GBuf;
__device__ __forceinline__ GBuf make_gbuf(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void glds16(GBuf b, uint32_t voff, char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(b, (__attribute__((address_space(3))) void*)lds_wave_base, 16,
                                             (int)voff, 0, 0, 0);
}
#endif
#ifndef MC_EMU
__device__ __forceinline__ half8_t gbuf_ld8(GBuf b, uint32_t voff) {
    return __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(b, (int)voff, 0, 0));
}
#endif
// 16-byte buffer store with the hardware range check (offset at / beyond the buffer's size: nothing is written)
#ifdef MC_EMU
__device__ inline void gbuf_st8(GBuf b, uint32_t voff, half8_t v) {
    if ((uint64_t)voff + 16 <= b.bytes) memcpy(const_cast<char*>(b.base) + voff, &v, 16);
}
// make this wave's earlier LDS writes visible to its own later LDS reads (same-wave LDS operations execute in order on the
// hardware; the simulator runs lanes as fibers, so every lane has to reach this point first)
__device__ inline void wave_lds_sync() { (void)hipemu::shfl(0, 0); }
#else
__device__ __forceinline__ void gbuf_st8(GBuf b, uint32_t voff, half8_t v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), b, (int)voff, 0, 0);
}
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#endif
// counted wait on this wave's outstanding vector-memory operations, and a bare workgroup barrier (no fence)
#ifdef MC_EMU
template <int N>
__device__ inline void wait_vmcnt_le() {}
__device__ inline void raw_barrier() { __syncthreads(); }
#else
template <int N>
__device__ __forceinline__ void wait_vmcnt_le() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void raw_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
#endif
// c + a.x b.x + a.y b.y with fp32 products and accumulation (v_dot2_f32_f16): row sums / sums of squares of fp16 data
#ifdef MC_EMU
__device__ inline float dot2acc(half2_t a, half2_t b, float c) { return c + (float)a[0] * (float)b[0] + (float)a[1] * (float)b[1]; }
#else
__device__ __forceinline__ float dot2acc(half2_t a, half2_t b, float c) { return __builtin_amdgcn_fdot2(a, b, c, false); }
#endif
// Identity the optimiser cannot see through: address arithmetic derived from the result is recomputed where it is used
// instead of being hoisted out of a long loop and kept (or spilled) in VGPRs for its whole duration.
#ifdef MC_EMU
__device__ inline int opaque(int x) { return x; }
__device__ inline float opaque(float x) { return x; }
#else
__device__ __forceinline__ float opaque(float x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}
#endif
// A voffset beyond every buffer we describe (callers keep buffers <= 2 GiB): forces the zero fill without any
// 32-bit wrap in the range check.
constexpr uint32_t kOOB = 0x80000000u;

// true iff the predicate holds on all 64 lanes (wave-uniform result)
#ifdef MC_EMU
__device__ inline bool wave_all(bool p) {
    int v = p ? 1 : 0;
    for (int m = 32; m >= 1; m >>= 1) v &= hipemu::shfl_xor(v, m);
    return v != 0;
}
__device__ inline bool wave_any(bool p) { return !wave_all(!p); }
__device__ inline float fast_exp2(float x) { return exp2f(x); }
#else
__device__ __forceinline__ bool wave_all(bool p) { return __all(p); }
__device__ __forceinline__ bool wave_any(bool p) { return __any(p); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
#endif

// hardware transpose read (ds_read_b64_tr_b16): within each 16-lane group lane i passes the address of halfs [4 (i & 3), +4)
// of row i >> 2 of a 4 x 16 block (any row pitch, 8-byte aligned) and receives column i of the block, rows 0..3 - a
// column of a row-major LDS image as an MFMA operand without a transposed copy
#ifdef MC_EMU
__device__ inline half4_t lds_read_tr4(const half_t* p) { return hipemu::lds_read_tr16_b64(p); }
#else
__device__ __forceinline__ half4_t lds_read_tr4(const half_t* p) {
    typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
    return __builtin_bit_cast(half4_t, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)p));
}
#endif
// The same read issued BEHIND THE COMPILER'S BACK (inline asm; `base` + OFF bytes): hipcc puts an s_waitcnt vmcnt(0) in front of
// the builtin form whenever an LDS-DMA load is in flight (it cannot tell the images apart), which drains a prefetch ring.  The
// result may only be used after lds_tr_wait() and a lds_tr_use() on the value.
#ifdef MC_EMU
template <int OFF>
__device__ inline half4_t lds_read_tr4_async(const char* base) {
    return hipemu::lds_read_tr16_b64(base + OFF);
}
__device__ inline void lds_tr_wait() {}
__device__ inline void lds_tr_use(half8_t&) {}
#else
template <int OFF>
__device__ __forceinline__ half4_t lds_read_tr4_async(const char* base) {
    half4_t r;
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)base;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF));
    return r;
}
__device__ __forceinline__ void lds_tr_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_tr_use(half8_t& v) { asm volatile("" : "+v"(v)); }   // orders the uses behind the wait
#endif
// max / sum over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48), every lane gets the result: two VALU lane swaps
// (v_permlane16_swap, v_permlane32_swap) instead of two LDS round trips (ds_bpermute)
#ifdef MC_EMU
__device__ inline float rows_max(float v) {
    unsigned x = __builtin_bit_cast(unsigned, v);
    auto r = hipemu::permlane_swap(x, x, 16);
    v = fmaxf(__builtin_bit_cast(float, r.a), __builtin_bit_cast(float, r.b));
    x = __builtin_bit_cast(unsigned, v);
    r = hipemu::permlane_swap(x, x, 32);
    return fmaxf(__builtin_bit_cast(float, r.a), __builtin_bit_cast(float, r.b));
}
__device__ inline void scale_in_place(f32x4& x, float a) { x *= a; }
__device__ inline float rows_sum(float v) {
    unsigned x = __builtin_bit_cast(unsigned, v);
    auto r = hipemu::permlane_swap(x, x, 16);
    v = __builtin_bit_cast(float, r.a) + __builtin_bit_cast(float, r.b);
    x = __builtin_bit_cast(unsigned, v);
    r = hipemu::permlane_swap(x, x, 32);
    return __builtin_bit_cast(float, r.a) + __builtin_bit_cast(float, r.b);
}
#else
__device__ __forceinline__ float rows_max(float v) {
    // (inline v_max: fmaxf on a bit-cast value costs a canonicalising v_max per operand)
    unsigned x = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    float m;
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(r[0]), "v"(r[1]));
    x = __builtin_bit_cast(unsigned, m);
    r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(r[0]), "v"(r[1]));
    return m;
}
__device__ __forceinline__ float rows_sum(float v) {
    // (inline v_add as well: hipcc folded r[0] + r[1] of the swap builtin into 2 * r[0])
    unsigned x = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    float m;
    asm("v_add_f32 %0, %1, %2" : "=v"(m) : "v"(r[0]), "v"(r[1]));
    x = __builtin_bit_cast(unsigned, m);
    r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    asm("v_add_f32 %0, %1, %2" : "=v"(m) : "v"(r[0]), "v"(r[1]));
    return m;
}
// x *= a in place (a tied inline v_mul): inside a rarely taken branch this keeps the untouched path free of register copies.
// ROUND 5 BUG FIX: `a` is the result of a transcendental (v_exp_f32) at the call site, and on gfx940+ a VALU instruction that
// reads a TRANS result needs one wait state in between (the "trans forwarding" hazard).  hipcc inserts that s_nop for its own
// instructions but does not look inside inline asm: the FIRST v_mul of a re-base read a stale register on some lanes, i.e. element 0
// of the first output tile of every re-based query row was scaled by whatever the register held before (the row offset): output
// channels 0 / 4 / 8 / 12 of a head wrong by a factor of 2 - 5 whenever a row's scores outgrew their first tile's maximum by more
// than 2^8 - never at the first key tile (the accumulators are zero there), rarely on N(0, 1) operands, on every real attention
// map with a wide logit range.  Found by tests/test_fullsize_parity.py::test_outlier_channels_stress; regression test:
// tests/test_kernels.py::test_attention_rows_that_outgrow_their_first_tile (GPU only: the simulator has no such hazard).  The
// s_nop is part of the FIRST multiply's asm statement, so nothing can be scheduled between the two.
__device__ __forceinline__ void scale_in_place(f32x4& x, float a) {
    float e0 = x[0];
#ifdef MC_CONTROL_TRANS_HAZARD   // TEST ONLY: the pre-fix code, built into tools/_build/libattn_trans_hazard.so as the negative control
    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(e0) : "v"(a));
#else
    asm volatile("s_nop 1\n\tv_mul_f32 %0, %0, %1" : "+v"(e0) : "v"(a));
#endif
    x[0] = e0;
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        float e = x[i];
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(e) : "v"(a));
        x[i] = e;
    }
}
#endif

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a compile-time constant in the body
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

// reductions across the 64 lanes of a wave
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor(v, m);
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, shfl_xor(v, m));
    return v;
}

__device__ inline float silu_f(float x) { return x / (1.0f + fast_exp(-x)); }
// d/dx silu(x) = s + x*s*(1-s), s = sigmoid(x)
__device__ inline float silu_grad_f(float x) {
    float s = 1.0f / (1.0f + fast_exp(-x));
    return s * (1.0f + x * (1.0f - s));
}
// erf via Abramowitz & Stegun 7.1.26 (|abs err| < 1.5e-7, far below fp16 resolution): one exp and a degree-5
// polynomial instead of the ~40-instruction libm erff - the exact-erf GELU of diffusers' GEGLU sits in GEMM epilogues.
__device__ inline float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = 1.0f / (1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.0f - poly * fast_exp(-ax * ax);
    return x < 0.f ? -r : r;
}
__device__ inline float gelu_f(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f)); }
__device__ inline float gelu_grad_f(float x) {
    float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752f));
    float pdf = 0.39894228040143268f * fast_exp(-0.5f * x * x);
    return cdf + x * pdf;
}

__device__ inline half8_t ld8(const half_t* p) { return *reinterpret_cast<const half8_t*>(p); }
__device__ inline void st8(half_t* p, half8_t v) { *reinterpret_cast<half8_t*>(p) = v; }
__device__ inline half4_t ld4(const half_t* p) { return *reinterpret_cast<const half4_t*>(p); }
__device__ inline void st4(half_t* p, half4_t v) { *reinterpret_cast<half4_t*>(p) = v; }
__device__ inline half8_t zero8() { half8_t z; for (int i = 0; i < 8; ++i) z[i] = (half_t)0.0f; return z; }
__device__ inline half4_t zero4() { half4_t z; for (int i = 0; i < 4; ++i) z[i] = (half_t)0.0f; return z; }
__device__ inline f32x4 fzero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

// Two K=16 MFMA steps as one K=32 instruction: the contraction index may be permuted freely as long as both
// operands use the same permutation, so the two half4 fragments of each operand are simply concatenated
// (k-slots 0-3 of every lane group = first step, 4-7 = second step).
__device__ inline half8_t cat4(half4_t lo, half4_t hi) {
    half8_t r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r[i] = lo[i];
        r[4 + i] = hi[i];
    }
    return r;
}

// two floats -> packed halfs, round toward zero (one v_cvt_pkrtz_f16_f32 instead of two converts and a pack)
#ifdef MC_EMU
__device__ inline half2_t pk_rtz(float a, float b) {
    half2_t r;
    float in[2] = {a, b};
    for (int i = 0; i < 2; ++i) {
        half_t h = (half_t)in[i];
        if (fabsf((float)h) > fabsf(in[i])) {  // rounded away from zero: step the magnitude back by one ulp
            uint16_t bits;
            __builtin_memcpy(&bits, &h, 2);
            bits -= 1;
            __builtin_memcpy(&h, &bits, 2);
        }
        r[i] = h;
    }
    return r;
}
#else
__device__ __forceinline__ half2_t pk_rtz(float a, float b) {
    return __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(a, b));
}
#endif
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// saturating float -> half (fp16 max 65504); keeps NaN out of downstream tensors on overflow.  One v_med3_f32: the
// fminf/fmaxf form costs an extra canonicalising v_max_f32 per value, which showed up in the GEMM epilogues.
#ifdef MC_EMU
__device__ inline half_t to_half(float x) {
    x = fminf(fmaxf(x, -65504.0f), 65504.0f);
    return (half_t)x;
}
#else
__device__ __forceinline__ half_t to_half(float x) {
    return (half_t)__builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f);
}
#endif

}  // namespace mc

// status codes returned through the C ABI (include/mc_kernels.h)
#define MC_OK 0
#define MC_ERR_SHAPE -1
#define MC_ERR_UNSUPPORTED -2
#define MC_ERR_LAUNCH -3
