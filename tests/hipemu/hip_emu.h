// TEST INFRASTRUCTURE ONLY — never linked into the product library.
//
// A small lane-accurate functional simulator for the HIP kernels under
// motionclone_amd/csrc.  The authoring container has no GPU, so the kernel
// sources are additionally compiled for the host with -DMC_EMU against this
// header; each GPU thread becomes a user-space fiber, a workgroup is scheduled
// round-robin on one OS thread, and the wave-level primitives the kernels use
// (64-lane shuffles, the two MFMA shapes with their gfx950 register layouts,
// __syncthreads) are reproduced with the documented lane <-> element mappings.
// It checks index math, masking and barrier placement; it says nothing about
// speed and is not a CPU fallback: the product loader (motionclone_amd/lib.py)
// only ever opens the hipcc-built gfx950 library.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
using std::max;
using std::min;

extern "C" void hipemu_switch(void** save_sp, void* load_sp);

namespace hipemu {

constexpr int kStackBytes = 96 * 1024;
constexpr int kSlotBytes = 128;  // per-lane exchange slot (largest: MFMA a+b fragments)

struct Fiber {
    void* sp = nullptr;
    dim3 tid;
    bool done = false;
    unsigned ops = 0;  // wave-collective counter (parity selects the exchange buffer)
};

struct WaveBuf {
    alignas(16) unsigned char slot[2][64][kSlotBytes];
};

struct BlockCtx {
    dim3 bid, bdim, gdim;
    Fiber* fibers = nullptr;
    int nfib = 0;
    int cur = 0;
    int live = 0;
    void* sched_sp = nullptr;
    int bar_arrived = 0;
    unsigned bar_gen = 0;
    char* dyn = nullptr;
    WaveBuf* waves = nullptr;
    const std::function<void()>* body = nullptr;
};

extern thread_local BlockCtx* g_blk;

inline void yield() {
    BlockCtx* b = g_blk;
    hipemu_switch(&b->fibers[b->cur].sp, b->sched_sp);
}

inline char* dyn_smem() { return g_blk->dyn; }

inline void syncthreads() {
    BlockCtx* b = g_blk;
    unsigned gen = b->bar_gen;
    b->bar_arrived++;
    for (;;) {
        if (b->bar_gen != gen) return;
        if (b->bar_arrived >= b->live) {  // last arriver (or peers exited): release
            b->bar_arrived = 0;
            b->bar_gen++;
            return;
        }
        yield();
    }
}

inline int lane_id() {
    BlockCtx* b = g_blk;
    const dim3& t = b->fibers[b->cur].tid;
    unsigned lin = t.x + b->bdim.x * (t.y + b->bdim.y * t.z);
    return (int)(lin & 63);
}
inline int wave_id() {
    BlockCtx* b = g_blk;
    const dim3& t = b->fibers[b->cur].tid;
    unsigned lin = t.x + b->bdim.x * (t.y + b->bdim.y * t.z);
    return (int)(lin >> 6);
}

// publish `mine` to the wave, let every other lane reach the same collective, return the buffer
inline unsigned char (*exchange(const void* mine, size_t bytes))[kSlotBytes] {
    BlockCtx* b = g_blk;
    Fiber& f = b->fibers[b->cur];
    int par = (int)(f.ops++ & 1u);
    WaveBuf& wb = b->waves[wave_id()];
    std::memcpy(wb.slot[par][lane_id()], mine, bytes);
    yield();
    return wb.slot[par];
}

template <class T>
inline T shfl(T v, int src) {
    auto buf = exchange(&v, sizeof(T));
    T r;
    std::memcpy(&r, buf[src & 63], sizeof(T));
    return r;
}
template <class T>
inline T shfl_xor(T v, int mask) {
    return shfl(v, lane_id() ^ mask);
}

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// v_mfma_f32_16x16x16_f16: lane l holds A[l&15][4*(l>>4)+j], B[4*(l>>4)+j][l&15];
// D[4*(l>>4)+i][l&15].
inline f4 mfma_16x16x16(h4 a, h4 b, f4 c) {
    struct { h4 a, b; } mine{a, b};
    auto buf = exchange(&mine, sizeof(mine));
    int l = lane_id();
    int col = l & 15, g = l >> 4;
    for (int i = 0; i < 4; ++i) {
        int row = 4 * g + i;
        float acc = c[i];
        for (int k = 0; k < 16; ++k) {
            decltype(mine) A, B;
            std::memcpy(&A, buf[row + 16 * (k >> 2)], sizeof(A));
            std::memcpy(&B, buf[col + 16 * (k >> 2)], sizeof(B));
            acc += (float)A.a[k & 3] * (float)B.b[k & 3];
        }
        c[i] = acc;
    }
    return c;
}

// v_mfma_f32_16x16x32_f16: lane l holds A[l&15][8*(l>>4)+j], B[8*(l>>4)+j][l&15]; D[4*(l>>4)+i][l&15].
inline f4 mfma_16x16x32(h8 a, h8 b, f4 c) {
    struct { h8 a, b; } mine{a, b};
    auto buf = exchange(&mine, sizeof(mine));
    int l = lane_id();
    int col = l & 15, g = l >> 4;
    for (int i = 0; i < 4; ++i) {
        int row = 4 * g + i;
        float acc = c[i];
        for (int k = 0; k < 32; ++k) {
            decltype(mine) A, B;
            std::memcpy(&A, buf[row + 16 * (k >> 3)], sizeof(A));
            std::memcpy(&B, buf[col + 16 * (k >> 3)], sizeof(B));
            acc += (float)A.a[k & 7] * (float)B.b[k & 7];
        }
        c[i] = acc;
    }
    return c;
}

// v_mfma_f32_32x32x16_f16: lane l holds A[l&31][8*(l>>5)+j], B[8*(l>>5)+j][l&31];
// D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
inline f16v mfma_32x32x16(h8 a, h8 b, f16v c) {
    struct { h8 a, b; } mine{a, b};
    auto buf = exchange(&mine, sizeof(mine));
    int l = lane_id();
    int col = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            decltype(mine) A, B;
            std::memcpy(&A, buf[row + 32 * (k >> 3)], sizeof(A));
            std::memcpy(&B, buf[col + 32 * (k >> 3)], sizeof(B));
            acc += (float)A.a[k & 7] * (float)B.b[k & 7];
        }
        c[r] = acc;
    }
    return c;
}

// ds_read_b64_tr_b16: within each 16-lane group lane i supplies the address of 4 contiguous 16-bit elements - columns
// [4 (i & 3), +4) of row i >> 2 of a 4 x 16 block - and receives column i of that block (rows 0..3).
inline h4 lds_read_tr16_b64(const void* p) {
    auto buf = exchange(&p, sizeof(p));
    int l = lane_id();
    int base = l & ~15, i = l & 15;
    h4 r;
    for (int j = 0; j < 4; ++j) {
        const void* src;
        std::memcpy(&src, buf[base + 4 * j + (i >> 2)], sizeof(src));
        r[j] = reinterpret_cast<const _Float16*>(src)[i & 3];
    }
    return r;
}

// v_permlane16_swap_b32 / v_permlane32_swap_b32 (rows = 16 lanes): the odd rows of `a` trade places with the even rows of `b`
// (16), the upper 32 lanes of `a` with the lower 32 lanes of `b` (32).  Returns the new (a, b).
struct u2 { unsigned a, b; };
inline u2 permlane_swap(unsigned a, unsigned b, int width) {
    u2 mine{a, b};
    auto buf = exchange(&mine, sizeof(mine));
    int l = lane_id();
    bool first_half = width == 16 ? ((l >> 4) & 1) == 0 : l < 32;
    u2 other;
    std::memcpy(&other, buf[l ^ width], sizeof(other));
    // lanes in the "even" half: a keeps, b receives the partner's a; lanes in the "odd" half: a receives the partner's b
    return first_half ? u2{a, other.a} : u2{other.b, b};
}

void run_block(BlockCtx& ctx, char* stacks);

template <class F>
inline void launch(dim3 grid, dim3 block, size_t smem, F&& fn) {
    std::function<void()> body = fn;
    const long nblocks = (long)grid.x * grid.y * grid.z;
    const int nthr = (int)(block.x * block.y * block.z);
    if (nblocks <= 0 || nthr <= 0) return;
    unsigned hw = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("MC_EMU_THREADS")) hw = (unsigned)std::atoi(e);
    if (hw < 1) hw = 1;
    int nworkers = (int)std::min<long>(hw, nblocks);
    std::atomic<long> next{0};
    auto worker = [&]() {
        char* stacks = (char*)std::malloc((size_t)nthr * kStackBytes);
        std::vector<Fiber> fibers(nthr);
        int nwaves = (nthr + 63) / 64;
        WaveBuf* waves = (WaveBuf*)std::aligned_alloc(16, sizeof(WaveBuf) * nwaves);
        char* dyn = smem ? (char*)std::aligned_alloc(16, (smem + 15) / 16 * 16) : nullptr;
        for (;;) {
            long bi = next.fetch_add(1);
            if (bi >= nblocks) break;
            BlockCtx ctx;
            ctx.bid = dim3((unsigned)(bi % grid.x), (unsigned)((bi / grid.x) % grid.y),
                           (unsigned)(bi / ((long)grid.x * grid.y)));
            ctx.bdim = block;
            ctx.gdim = grid;
            ctx.fibers = fibers.data();
            ctx.nfib = nthr;
            ctx.dyn = dyn;
            ctx.waves = waves;
            ctx.body = &body;
            run_block(ctx, stacks);
        }
        std::free(stacks);
        std::free(waves);
        if (dyn) std::free(dyn);
    };
    if (nworkers == 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (int i = 0; i < nworkers; ++i) pool.emplace_back(worker);
        for (auto& t : pool) t.join();
    }
}

}  // namespace hipemu

#define threadIdx (hipemu::g_blk->fibers[hipemu::g_blk->cur].tid)
#define blockIdx (hipemu::g_blk->bid)
#define blockDim (hipemu::g_blk->bdim)
#define gridDim (hipemu::g_blk->gdim)
#define __syncthreads() hipemu::syncthreads()

inline float atomicAdd(float* p, float v) {
    // blocks run concurrently on host threads
    std::atomic_ref<float> r(*p);
    float old = r.load();
    while (!r.compare_exchange_weak(old, old + v)) {
    }
    return old;
}
inline uint32_t atomicAdd(uint32_t* p, uint32_t v) {
    std::atomic_ref<uint32_t> r(*p);
    return r.fetch_add(v);
}
