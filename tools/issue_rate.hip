// Issue-rate probes for gfx950 (tools only): how many cycles a wave64 VALU / transcendental / MFMA instruction occupies its
// pipe, and whether VALU of one wave overlaps MFMAs of ANOTHER wave on the same SIMD, or of the SAME wave.
// One workgroup per CU, 4 or 8 waves (one or two per SIMD); waves 0-3 run kind_a, waves 4-7 kind_b; `iters` trips of a body of 64
// independent instructions; s_memtime around the loop, per wave.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define REP8(x) x x x x x x x x
#define REGS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define FMA_BODY asm volatile(REP8("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n") : REGS : "v"(c));
#define MUL_BODY asm volatile(REP8("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n") : REGS : "v"(c));
#define EXP_BODY asm volatile(REP8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n") : REGS);
#define CVT_BODY asm volatile(REP8("v_cvt_pkrtz_f16_f32 %0, %1, %2\n v_cvt_pkrtz_f16_f32 %3, %4, %5\n v_cvt_pkrtz_f16_f32 %6, %7, %0\n v_cvt_pkrtz_f16_f32 %1, %2, %3\n v_cvt_pkrtz_f16_f32 %4, %5, %6\n v_cvt_pkrtz_f16_f32 %7, %0, %1\n v_cvt_pkrtz_f16_f32 %2, %3, %4\n v_cvt_pkrtz_f16_f32 %5, %6, %7\n") : REGS);
#define MAX3_BODY asm volatile(REP8("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %3, %3, %4, %5\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %7, %7, %0, %1\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %5, %5, %6, %7\n") : REGS);
#define EXPF16_BODY asm volatile(REP8("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7\n") : REGS);
#define PKFMA16_BODY asm volatile(REP8("v_pk_fma_f16 %0, %0, %8, %8\n v_pk_fma_f16 %1, %1, %8, %8\n v_pk_fma_f16 %2, %2, %8, %8\n v_pk_fma_f16 %3, %3, %8, %8\n v_pk_fma_f16 %4, %4, %8, %8\n v_pk_fma_f16 %5, %5, %8, %8\n v_pk_fma_f16 %6, %6, %8, %8\n v_pk_fma_f16 %7, %7, %8, %8\n") : REGS : "v"(c));
#define PKMUL32_BODY asm volatile(REP8("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p4));
// 16 MFMAs on 4 accumulators (each dependent on the one 4 back)
#define MFMA16                                                                   \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                              \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc0, 0, 0, 0);    \
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc1, 0, 0, 0);    \
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc2, 0, 0, 0);    \
        acc3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc3, 0, 0, 0);    \
    }
#define TIMED(body)                                    \
    {                                                  \
        t0 = __builtin_readcyclecounter();             \
        for (int it = 0; it < iters; ++it) { body }    \
        t1 = __builtin_readcyclecounter();             \
    }

// kinds: 0 idle, 1 fma, 2 exp f32, 3 cvt_pkrtz, 4 max3, 5 mul, 6 exp f16, 7 pk_fma_f16, 8 pk_mul_f32, 10 MFMA (64 per trip),
// 21 same wave: 16 MFMA + 64 mul per trip, 22 same wave: 16 MFMA + 64 exp, 23: 16 MFMA + 32 mul + 32 exp
extern "C" __global__ __launch_bounds__(512) void probe(int kind_a, int kind_b, int iters, uint64_t* cycles, float* sink) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kind = wave < 4 ? kind_a : kind_b;
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {0.999f, 0.999f};
    const float c = 0.999f;
    h8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.01f * i); hb[i] = (_Float16)(0.02f * i); }
    f4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    uint64_t t0 = 0, t1 = 0;
    __syncthreads();
    if (kind == 1) TIMED(FMA_BODY)
    else if (kind == 2) TIMED(EXP_BODY)
    else if (kind == 3) TIMED(CVT_BODY)
    else if (kind == 4) TIMED(MAX3_BODY)
    else if (kind == 5) TIMED(MUL_BODY)
    else if (kind == 6) TIMED(EXPF16_BODY)
    else if (kind == 7) TIMED(PKFMA16_BODY)
    else if (kind == 8) TIMED(PKMUL32_BODY)
    else if (kind == 10) TIMED(MFMA16 MFMA16 MFMA16 MFMA16)
    else if (kind == 21) TIMED(MFMA16 MUL_BODY)
    else if (kind == 22) TIMED(MFMA16 EXP_BODY)
    else if (kind == 23) TIMED(MFMA16 asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n" : REGS : "v"(c)); MFMA16)
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc0[0] + acc1[1] + acc2[2] + acc3[3] + p0[0] + p1[1] + p2[0] + p3[1];
}

extern "C" int run_probe(int kind_a, int kind_b, int nwaves, int iters, uint64_t* host_cycles /* [8] */, float* ms) {
    uint64_t* dcy;
    float* dsink;
    const int blocks = 256;
    if (hipMalloc(&dcy, blocks * 8 * sizeof(uint64_t)) != hipSuccess) return -1;
    if (hipMalloc(&dsink, blocks * 512 * sizeof(float)) != hipSuccess) return -1;
    hipMemset(dcy, 0, blocks * 8 * sizeof(uint64_t));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(64 * nwaves), 0, 0, kind_a, kind_b, 10, dcy, dsink);   // warm
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(64 * nwaves), 0, 0, kind_a, kind_b, iters, dcy, dsink);
    hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    hipEventElapsedTime(ms, e0, e1);
    uint64_t all[8];
    hipMemcpy(all, dcy, sizeof(all), hipMemcpyDeviceToHost);   // block 0
    for (int i = 0; i < 8; ++i) host_cycles[i] = all[i];
    hipFree(dcy);
    hipFree(dsink);
    return 0;
}
