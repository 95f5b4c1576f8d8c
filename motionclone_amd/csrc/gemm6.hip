// Persistent tile loop over gemm5's 256x320 tile (round 6): ONE 8-wave workgroup per CU walks a list of output tiles and
// the operand ring never stops at a tile boundary.
//
// Why (profiles/r05_vendor_anchor.md): on the wide-N short-K Linear layers (FeedForward's first Linear,
// q|k|v of the 1280-channel level: attention.py:211,288,355-357, motion_module.py:209,222) gemm5's k-loop already runs at
// the vendor library's rate, but every 256x320 tile pays ~8.5 us that is not k-loop - ~5.5 us of it the prologue: all
// 256 workgroups of a dispatch wave fill three ring stages at the same moment, and nothing is multiplied until the first
// one lands.  Here a workgroup starts that burst once per LAUNCH:
//
//   * flat stage stream: stage g of the workgroup's work list belongs to tile g / nk; the LDS-DMA loads of stage g + 3
//     are issued while stage g is multiplied, whichever tile they belong to, so the first three stages of tile t + 1 land
//     under the last three k-steps and the epilogue of tile t;
//   * the epilogue keeps the ring: its fp16 images are 32 rows x 80 columns per wave (5.5 KiB; gemm5: 32 x 160) and live in
//     the ONE ring slot that is free at a tile boundary (waves 0-5) plus 11 KiB behind the ring (waves 6-7) - the three
//     prefetched stages of the next tile stay where they are.  A 64 x 160 wave tile leaves in four passes (two for the
//     fused GEGLU) instead of two;
//   * epilogue loads / stores go through buffer descriptors with the hardware range check, so a wave issues the SAME number
//     of vector-memory instructions for every tile (no predicated-off instruction): the counted s_waitcnt vmcnt of the two
//     stages that follow an epilogue allow exactly the epilogue's stores on top of the ring's loads (vmcnt retires in issue
//     order on gfx9: the ring of gemm5 already relies on it for LDS-DMA; VAR bit 0 builds the variant that does not extend
//     the assumption to stores - every wait as in the steady state, i.e. the stores are drained at the first k-step);
//   * tile order: XCD x owns the M-tiles tm = 8 i + x and walks them in super-tiles (gemm5's order: the activation rows of sm
//     tiles and the weight rows of sn tiles are what an XCD's 32 workgroups share in their 4 MiB L2).  STATIC: workgroup
//     (x, s) takes locals s, s + S, ... of its XCD's list.  DYNAMIC (ctr != null): the first local is s, every further one
//     comes from a per-XCD counter word (device-scope atomic, requested by one lane at the START of an epilogue and read at
//     its end - the compiler's own exact vmcnt wait, no drain of the ring - for the tile AFTER the one already being loaded,
//     handed to the other waves through LDS) - a workgroup that starts late because another launch sequence holds its CU
//     takes fewer tiles instead of finishing its whole list late.  The counter block (9 words) is zero when the kernel starts and the LAST workgroup to
//     leave zeroes it again (arrival count in word 8), so a caller hands the same zero-initialised words to every launch of a
//     stream (graph replays included) without a memset node.
//
// Output is bit-identical to gemm5's: every output element is one fp32 accumulation chain along k in the same order.
// DENSE only (the 3x3 convs of these levels fill the chip with one round of tiles and are not short-K).
#include "gemm5_tile.hpp"

namespace mc {

namespace g6 {
using T = g5::Tile<256>;
constexpr int IMG = 32 * g5::RSG;                // one wave's epilogue image: 32 rows x (80 columns + 16 B) = 5632 B
constexpr int IMG_IN_SLOT = 6;                   // waves 0..5 inside the free ring slot, 6..7 behind the ring
constexpr int BSTRIP = 640;                      // one wave's bias strip: 160 fp32
constexpr int TAIL = 2 * IMG + 4 * BSTRIP;       // behind the ring: images of waves 6-7, bias strips of waves 4-7
constexpr size_t SMEM = T::SMEM + TAIL + 16;     // ... and the tile hand-over word
constexpr int NK_MIN = 8;                        // stages per tile the schedule assumes at least (K >= 256)
constexpr int PMIN = 8;                          // stream-K: stages a piece of a tile has at least
constexpr int SLAB_FLOATS = 256 * 320;           // one workgroup's accumulators
constexpr int FLAG0 = 16;                        // first "partial sums ready" word of the counter block
constexpr int CTR_WORDS = 512;                   // counter block: 2 KiB
static_assert(IMG_IN_SLOT * IMG + 4 * BSTRIP <= T::STAGE && SMEM <= 160 * 1024, "epilogue images: one ring slot + the tail");

// local index `l` of XCD `xcd`'s tile list -> (tm, tn); false past the end.  Rows of the XCD: tm = 8 i + xcd, walked in groups
// of sm rows, inside a group in column blocks of sn tiles, rows fastest (gemm5's super-tile order without holes).
__device__ __forceinline__ int list_len(int xcd, int tilesM, int tilesN, int flat) {
    return flat == 1 ? (tilesM * tilesN - xcd + 7) >> 3 : ((tilesM - xcd + 7) >> 3) * tilesN;
}
__device__ __forceinline__ bool tile_of(int l, int xcd, int tilesM, int tilesN, int sm, int sn, int flat, int& tm, int& tn) {
    if (flat == 1) {
        const int f = 8 * l + xcd;
        if (l < 0 || f >= tilesM * tilesN) return false;
        tn = f / tilesM;
        tm = f - tn * tilesM;
        return true;
    }
    const int rows = (tilesM - xcd + 7) >> 3;
    if (l >= rows * tilesN) return false;
    if (flat == 2) {
        // COLUMN BLOCKS OUTERMOST (A/B, mc_gemm_tileloop_f16 flags 0x4): the sn weight panels of a column block stay in the XCD's L2
        // while ALL of its row groups pass (W crosses the fabric once per XCD, A once per column block) - the default keeps the sm
        // activation panels of a row group and streams W past them (A once, W once per row group)
        const int per_blk = rows * sn;
        const int blk = l / per_blk, idx = l - blk * per_blk;
        const int g = idx / (sm * sn), w = idx - g * (sm * sn);
        const int rg = min(sm, rows - g * sm);
        tn = blk * sn + w / rg;
        tm = (g * sm + w % rg) * 8 + xcd;
        return true;
    }
    const int per_group = sm * tilesN;
    const int g = l / per_group, idx = l - g * per_group;
    const int rg = min(sm, rows - g * sm);
    const int blk = idx / (rg * sn), w = idx - blk * (rg * sn);
    tn = blk * sn + w / rg;
    tm = (g * sm + w % rg) * 8 + xcd;
    return true;
}
// stream-K: first stage of range r (0 .. n) when the S stages of an XCD's list are dealt to n workgroups.  A boundary that would
// leave a piece of fewer than PMIN stages on either side of it moves to the tile boundary (the k-loop's look-ahead assumes
// PMIN stages per piece).
__device__ __forceinline__ int sk_bound(int r, int n, int S, int nk) {
    int b = (int)((long)r * S / n);
    const int rem = b % nk;
    if (rem < PMIN) b -= rem;
    else if (nk - rem < PMIN) b += nk - rem;
    return b;
}
}  // namespace g6

// Everything the kernel is told, in ONE by-value block: the parts only the epilogue / the tile switch read are fetched there
// through a laundered pointer (scalar loads at the point of use) instead of living in SGPRs across the k-loop, which has
// none to spare (the first build spilled 78 of them into VGPR lanes and 35 VGPRs to scratch).
struct G6Args {
    GemmParams p;
    uint32_t bytesA, bytesW, bytesC, bytesR, bytesB;
    int tilesM, tilesN, sm, sn;
    int mode;          // 0 static tile order, 1 dynamic (per-XCD counters), 2 stream-K (stage ranges, tiles may be cut along k)
    int flat;          // tile lists without super-tiles (fewer than 8 rows of tiles): list entry l of XCD x = tile 8 l + x, rows fastest
    uint32_t* ctr;     // words 0-7 tile counters, 8 arrival count, 16 + 32 x + r: "partial sums of range r of XCD x are in memory"
    float* slabs;      // stream-K: [8][32] fp32 accumulator images (256 x 320 each)
    uint32_t bytesS;
};
#ifdef MC_EMU
#define G6_ARGS (&args)      // simulator: the parameter is an ordinary object
typedef const G6Args* g6args_t;
__device__ inline g6args_t late(const G6Args* a) { return a; }
#else
// (the kernarg segment is constant memory: through an address_space(4) pointer held in SGPRs every field is an s_load and stays
// wave-uniform - through a laundered GENERIC pointer the fields came back as flat loads in VGPRs and every buffer instruction of
// the epilogue was wrapped in a waterfall loop)
#define G6_ARGS nullptr      // device: never take the parameter's address (see `late`)
typedef const G6Args __attribute__((address_space(4)))* g6args_t;
__device__ __forceinline__ g6args_t late(const G6Args*) {
    // the block is the kernel's only parameter: offset 0 of the kernarg segment.  (Taking the parameter's address instead
    // makes hipcc copy it to scratch and read EVERYTHING - the operand descriptors of the k-loop included - back through VGPRs.)
    uint64_t v = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(v));
    return (g6args_t)v;
}
#endif

// one 32-bit word of LDS, explicitly in the LDS address space: through a `volatile uint32_t*` the hand-over word became a FLAT
// access with s_waitcnt vmcnt(0) - a drain of the operand ring - in front of it
#ifdef MC_EMU
__device__ inline uint32_t lds_ld32(const char* p) { return *reinterpret_cast<const volatile uint32_t*>(p); }
__device__ inline void lds_st32(char* p, uint32_t v) { *reinterpret_cast<volatile uint32_t*>(p) = v; }
#else
__device__ __forceinline__ uint32_t lds_ld32(const char* p) {
    uint32_t v;
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}
__device__ __forceinline__ void lds_st32(char* p, uint32_t v) {
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)p;
    asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory");
}
#endif

// stream-K hand-over primitives: 16-byte buffer store / load that bypass the non-coherent levels (sc1: write-through / L2-served),
// and a device-scope flag
#ifdef MC_EMU
__device__ inline void gbuf_st16_wt(GBuf b, uint32_t voff, f32x4 v) { gbuf_st8(b, voff, __builtin_bit_cast(half8_t, v)); }
__device__ inline f32x4 gbuf_ld16_wt(GBuf b, uint32_t voff) { return __builtin_bit_cast(f32x4, gbuf_ld8(b, voff)); }
__device__ inline void flag_store(uint32_t* f, uint32_t v) { std::atomic_ref<uint32_t>(*f).store(v, std::memory_order_release); }
__device__ inline void flag_wait(uint32_t* f) {
    while (std::atomic_ref<uint32_t>(*f).load(std::memory_order_acquire) == 0) std::this_thread::yield();
}
#else
__device__ __forceinline__ void gbuf_st16_wt(GBuf b, uint32_t voff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), b, (int)voff, 0, 16);   // aux 16 = sc1
}
__device__ __forceinline__ f32x4 gbuf_ld16_wt(GBuf b, uint32_t voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b, (int)voff, 0, 16));
}
__device__ __forceinline__ void flag_store(uint32_t* f, uint32_t v) {
    __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void flag_wait(uint32_t* f) {
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(4);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
#endif

template <int EPI, int RES, int VAR, int SK>
__global__ __launch_bounds__(512, 1) void gemm6_kernel(G6Args args) {
    using g5::TN; using g5::BKT; using g5::RPI; using g5::lds_off32;
    using T = g6::T;
    constexpr int NW = g5::NW, NS = g5::NS, BM = 256, BN = g5::BN;
    constexpr int TM = T::TM, RA = T::RA, STAGE = T::STAGE, A_BYTES = T::A_BYTES, LA = T::LA, LB = T::LB, WB = T::WB, WX = T::WX;
    constexpr int WMW = T::WMW;
    constexpr int ST = TM * (EPI == 1 ? 1 : 2) * 6;   // buffer stores one epilogue issues per wave, always
    static_assert(NS == 4 && LA + ST + LA < 64, "vmcnt is a 6-bit counter");
    MC_DYN_SMEM(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef MC_EMU
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const bool grpA = wave < WX;
    const int xcd = blockIdx.x & 7, nslots = gridDim.x >> 3;
    const int nk = args.p.K / BKT;
    constexpr bool sk = SK != 0;     // stream-K is its own instantiation: its hand-over code costs the other modes registers
    const bool dynamic = !sk && args.mode == 1;
    // stream-K deals the ranges in REVERSE dispatch order: the workgroup that finishes a cut tile (it owns the tile's first
    // stages, at the END of its range) waits for workgroups with LOWER block indices, which were dispatched before it
    const int slot = sk ? nslots - 1 - (int)(blockIdx.x >> 3) : (int)(blockIdx.x >> 3);
    char* handover = smem + T::SMEM + g6::TAIL;

    const GBuf bufA = make_gbuf(args.p.A, args.bytesA);
    const GBuf bufW = make_gbuf(args.p.W, args.bytesW);

    // the last workgroup to leave zeroes the counter block for the next launch that is handed the same words
    auto leave = [&]() {
        if (dynamic && tid == 0) {
            uint32_t* ctr = late(G6_ARGS)->ctr;
            const uint32_t before = atomicAdd(ctr + 8, 1u);
            if (before == gridDim.x - 1) {
#pragma unroll
                for (int i = 0; i < 9; ++i) ctr[i] = 0u;
            }
        }
    };

    // ---- the tile whose operands are being LOADED (at most one tile ahead of the one being multiplied) -------------------
    int l_m0 = 0, l_n0 = 0;
    bool l_valid = false;
    uint32_t a_off[RA], w_off[WB + 1];   // byte offset of the lane's 16 bytes at k = 0, or kOOB: hardware zero fill
    auto setup_offsets = [&]() {     // the lane's operand offsets of the load tile (recomputed after every epilogue, see lane_now)
        const g6args_t q = late(G6_ARGS);
        const int ln = lane_now(), rsub = ln >> 2, lslot = (ln & 3) ^ (ln >> 4);
        const int M = q->p.M, N = q->p.N, lda = q->p.lda, K = q->p.K;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int m = l_m0 + (wave + NW * i) * RPI + rsub;
            a_off[i] = (l_valid && m < M) ? ((uint32_t)m * (uint32_t)lda + (uint32_t)lslot * 8u) * 2u : kOOB;
        }
#pragma unroll
        for (int i = 0; i <= WB; ++i) {
            const int n = l_n0 + (i < WB ? wave + NW * i : WB * NW + (wave % WX)) * RPI + rsub;
            w_off[i] = (l_valid && n < N) ? (uint32_t)n * (uint32_t)K * 2u + (uint32_t)lslot * 16u : kOOB;
        }
    };
    auto setup_tile = [&](int local) {   // scalar part
        const g6args_t q = late(G6_ARGS);
        int tm = 0, tn = 0;
        l_valid = g6::tile_of(local, xcd, q->tilesM, q->tilesN, q->sm, q->sn, q->flat, tm, tn);
        l_m0 = tm * BM;
        l_n0 = tn * BN;
    };
    // one LDS-DMA instruction of k-stage kt of the load tile into ring slot buf: pieces 0 .. RA-1 activation row groups,
    // then 2 (3) weight row groups (kOOB + a small offset stays out of range)
    auto issue_piece = [&](int kt, int buf, int piece) {
        char* base = smem + buf * STAGE;
        const uint32_t kb = (uint32_t)kt * (BKT * 2u);
        if (piece < RA) {
            glds16(bufA, a_off[piece] + kb, base + (wave + NW * piece) * 1024);
        } else {
            const int i = piece - RA;
            glds16(bufW, w_off[i] + kb, base + A_BYTES + (i < WB ? wave + NW * i : WB * NW + (wave % WX)) * 1024);
        }
    };
    // ---- work list ----------------------------------------------------------------------------------------------------------
    int t_load = slot;
    int t_next = slot + nslots;          // static order; dynamic: replaced by the counter's answers
    int lkt = 0, l_kend = nk;            // next k-stage of the load tile / end of the piece being loaded
    int r_end = 0;                       // stream-K: end of this workgroup's range, in stages of its XCD's list
    if (sk) {
        const g6args_t q = late(G6_ARGS);
        const int S = g6::list_len(xcd, q->tilesM, q->tilesN, q->flat) * nk;
        const int r_beg = g6::sk_bound(slot, nslots, S, nk);
        r_end = g6::sk_bound(slot + 1, nslots, S, nk);
        if (r_beg >= r_end) return;      // (nothing to reset: stream-K keeps no counters)
        t_load = r_beg / nk;
        lkt = r_beg - t_load * nk;
        l_kend = min(nk, r_end - t_load * nk);
    }
    setup_tile(t_load);
    if (!l_valid) {
        leave();
        return;
    }
    int c_kbeg = lkt, c_cnt = l_kend - lkt;   // the piece being multiplied: first k-stage, stages
    int n_cnt = nk;                           // ... and the stages of the one being loaded, once the load side has moved on
    // dynamic order: the second tile's index is requested now and handed over after the prologue's wait (which covers it:
    // vmcnt retires in issue order and the request is older than every operand load)
    uint32_t ticket = 0;
    if (dynamic && wave == 0 && lane == 0) ticket = atomicAdd(args.ctr + xcd, 1u);
    int c_m0 = l_m0, c_n0 = l_n0;        // the tile being multiplied

    f32x16 acc[TN][TM];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };

    const int wr = wave % WMW, wc = wave / WMW;
    const int wm0 = wr * (32 * TM), wn0 = wc * 160;

    half8_t fa[2][TM], fw[2][TN];
    int l31 = 0, lhi = 0;            // set right before their first use and again after every epilogue: nothing lane-derived
                                     // lives across an epilogue (see lane_now)
    auto load_frags = [&](int buf, int ks, half8_t* a, half8_t* w) {
        const char* bA = smem + buf * STAGE;
        const char* bW = bA + A_BYTES;
#pragma unroll
        for (int j = 0; j < TM; ++j)
            a[j] = *reinterpret_cast<const half8_t*>(bA + lds_off32(wm0 + 32 * j + l31, 2 * ks + lhi));
#pragma unroll
        for (int i = 0; i < TN; ++i)
            w[i] = *reinterpret_cast<const half8_t*>(bW + lds_off32(wn0 + 32 * i + l31, 2 * ks + lhi));
    };
    // half a stage, exactly gemm5's scheduling region: MFMAs of the current k-slice, fragment reads of the next one, NL LDS-DMA
    // instructions (pieces P0 ..) of load-tile stage `lk` into slot `lbuf`
    auto half_step = [&](auto nl_tag, auto p0_tag, int rbuf, int rks, half8_t* ra, half8_t* rw, const half8_t* ca, const half8_t* cw,
                         int lk, int lbuf) {
        constexpr int NL = decltype(nl_tag)::value;
        constexpr int P0 = decltype(p0_tag)::value;
        static_assert(NL <= TN, "one LDS-DMA behind each weight fragment at most");
        const char* bA = smem + rbuf * STAGE;
        const char* bW = bA + A_BYTES;
#pragma unroll
        for (int j = 0; j < TM; ++j)
            ra[j] = *reinterpret_cast<const half8_t*>(bA + lds_off32(wm0 + 32 * j + l31, 2 * rks + lhi));
#pragma unroll
        for (int i = 0; i < TN; ++i) {
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = mfma32(cw[i], ca[j], acc[i][j]);
            rw[i] = *reinterpret_cast<const half8_t*>(bW + lds_off32(wn0 + 32 * i + l31, 2 * rks + lhi));
            if (i < NL) issue_piece(lk, lbuf, P0 + i);
        }
#ifndef MC_EMU
        __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (i < NL) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
#endif
    };

    int buf = 0;
    auto nxt = [](int b) { return b + 1 == NS ? 0 : b + 1; };
    auto slot_of_new = [](int b) { return b == 0 ? NS - 1 : b - 1; };

    // the load side moves on to the next tile of the list (all waves, uniformly, at the top of a stage)
    auto advance = [&]() {
        if (dynamic) {
            const uint32_t v = lds_ld32(handover);
#ifdef MC_EMU
            t_next = (int)v;
#else
            t_next = __builtin_amdgcn_readfirstlane((int)v);
#endif
        }
        if (sk) {
            t_load += 1;                 // the next tile of the list, as far as the range reaches into it
            setup_tile(t_load);
            l_valid = l_valid && t_load * nk < r_end;
            l_kend = min(nk, r_end - t_load * nk);
        } else {
            t_load = t_next;
            setup_tile(t_load);
            t_next = t_load + nslots;    // (static order)
        }
        setup_offsets();
        lkt = 0;
        n_cnt = l_kend;
    };

    // ---- stream-K hand-over of partial sums --------------------------------------------------------------------------------
    // Slab of range r of XCD x: accumulator-native order, chunk (i, j, q) of lane l of wave w at float4 index
    // ((w 20 TM + (i TM + j) 4 + q) 64 + l: every instruction moves 1 KiB contiguously.  Producer: write-through (sc1) stores,
    // each wave waits for its own (vmcnt(0)), workgroup barrier, ONE lane stores the flag (device scope).  Consumer: ONE lane
    // polls the flag (relaxed, device scope) and runs the device-scope acquire, workgroup barrier, sc1 loads
    // (MI355X_MICROARCH.md: inter-workgroup visibility).  The consumer clears the flag: the block is zero again when the kernel ends.
    auto slab_voff = [&](int range) {
        return (uint32_t)((xcd * 32 + range) * (g6::SLAB_FLOATS / 4) + wave * (TN * TM * 4 * 64) + lane_now()) * 16u;
    };
    auto store_partial = [&]() {
        const g6args_t q = late(G6_ARGS);
        const GBuf bufS = make_gbuf(q->slabs, q->bytesS);
        const uint32_t base = slab_voff(slot);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * qq + e];
                    gbuf_st16_wt(bufS, base + (uint32_t)(((i * TM + j) * 4 + qq) * 64) * 16u, v);
                }
    };
    auto raise_flag = [&]() {
        if (wave == 0 && lane_now() == 0)
            flag_store(late(G6_ARGS)->ctr + g6::FLAG0 + xcd * 32 + slot, 1u);
    };
    auto gather_partials = [&](int k_have) {
        const g6args_t q = late(G6_ARGS);
        const GBuf bufS = make_gbuf(q->slabs, q->bytesS);
        const int S = g6::list_len(xcd, q->tilesM, q->tilesN, q->flat) * nk;
        const int tile_end = (r_end / nk + 1) * nk;       // this piece starts its tile: the tile is the one r_end lies in
        (void)k_have;
        for (int r = slot + 1; r < nslots && g6::sk_bound(r, nslots, S, nk) < tile_end; ++r) {
            uint32_t* flag = q->ctr + g6::FLAG0 + xcd * 32 + r;
            if (wave == 0 && lane_now() == 0) {
                flag_wait(flag);
                flag_store(flag, 0u);
            }
            raw_barrier();
            const uint32_t base = slab_voff(r);
            // (compile-time indices through static_for: with `#pragma unroll` loops inside this run-time loop hipcc kept the
            // accumulator array in scratch memory)
            static_for<TN>([&](auto ic) {
                static_for<TM>([&](auto jc) {
                    constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
                    const f32x4 v0 = gbuf_ld16_wt(bufS, base + (uint32_t)(((i * TM + j) * 4 + 0) * 64) * 16u);
                    const f32x4 v1 = gbuf_ld16_wt(bufS, base + (uint32_t)(((i * TM + j) * 4 + 1) * 64) * 16u);
                    const f32x4 v2 = gbuf_ld16_wt(bufS, base + (uint32_t)(((i * TM + j) * 4 + 2) * 64) * 16u);
                    const f32x4 v3 = gbuf_ld16_wt(bufS, base + (uint32_t)(((i * TM + j) * 4 + 3) * 64) * 16u);
                    acc[i][j][0] += v0[0]; acc[i][j][1] += v0[1]; acc[i][j][2] += v0[2]; acc[i][j][3] += v0[3];
                    acc[i][j][4] += v1[0]; acc[i][j][5] += v1[1]; acc[i][j][6] += v1[2]; acc[i][j][7] += v1[3];
                    acc[i][j][8] += v2[0]; acc[i][j][9] += v2[1]; acc[i][j][10] += v2[2]; acc[i][j][11] += v2[3];
                    acc[i][j][12] += v3[0]; acc[i][j][13] += v3[1]; acc[i][j][14] += v3[2]; acc[i][j][15] += v3[3];
                });
            });
        }
    };

    auto epi_args = [&]() {          // fetched where the epilogue runs (see `late`)
        const g6args_t q = late(G6_ARGS);
        EpiArgs e;
        e.C = q->p.C; e.R = q->p.R; e.M = q->p.M; e.N = q->p.N; e.ldc = q->p.ldc; e.ldr = q->p.ldr; e.alpha = q->p.alpha;
        e.bytesC = q->bytesC; e.bytesR = q->bytesR; e.has_bias = q->p.bias != nullptr;
        return e;
    };

    auto run = [&](auto ga_tag) {
        constexpr bool GA = decltype(ga_tag)::value;
        constexpr int L = GA ? LA : LB;
        constexpr int P1 = L < TN ? L : TN, P2 = L - P1;
        using N1 = std::integral_constant<int, P1>;
        using N2 = std::integral_constant<int, P2>;
        using N0 = std::integral_constant<int, 0>;
        // one k-stage of the compute tile; the loads it issues are stage lkt of the load tile.  EXTRA: vector-memory
        // instructions younger than the ring stage this wait is about, besides the ring's own (the epilogue's stores)
        auto stage = [&](auto extra_tag) {
            constexpr int EXTRA = decltype(extra_tag)::value;
            const int nbuf = nxt(buf), lbuf = slot_of_new(buf);
            half_step(N1(), N0(), buf, 1, fa[1], fw[1], fa[0], fw[0], lkt, lbuf);
            wait_vmcnt_le<L + P1 + EXTRA>();
            raw_barrier();
            half_step(N2(), N1(), nbuf, 0, fa[0], fw[0], fa[1], fw[1], lkt, lbuf);
            buf = nbuf;
            ++lkt;
        };
        using X0 = std::integral_constant<int, 0>;
        using XS = std::integral_constant<int, (VAR & 1) ? 0 : ST>;
        // Prologue, once per launch, INSIDE this wave group's copy of the code: whatever is computed per lane in front of
        // the group branch and used behind it would be spilled across the other group's copy, and a value that is reloaded
        // from scratch in front of a loop costs an s_waitcnt vmcnt(0) at the loop's header in EVERY iteration (hipcc merges
        // the header's pending-load state over both incoming edges).
        setup_offsets();
        zero_acc();
#pragma unroll
        for (int s0 = 0; s0 < NS - 1; ++s0) {
#pragma unroll
            for (int q = 0; q < L; ++q) issue_piece(lkt, s0, q);
            ++lkt;
        }
        // stage 0 of the first tile landed: own loads (two younger stages stay in flight), then everybody's
        wait_vmcnt_le<2 * L>();
        if (dynamic && wave == 0 && lane_now() == 0) lds_st32(handover, (uint32_t)nslots + ticket);
        raw_barrier();
        {
            const int ln = lane_now();
            l31 = ln & 31;
            lhi = ln >> 5;
        }
        load_frags(0, 0, fa[0], fw[0]);
        int k = 0;
        bool finish = false;
        for (;;) {
            for (; k < c_cnt - 1; ++k) {
                if (lkt == l_kend) advance();
                stage(X0());
            }
            // the tile's last stage, with the wave's 160 bias values requested in front of it (one more load in the queue:
            // the stage's wait only becomes stricter, and the epilogue's wait for it is an exact count of the ring pieces)
            f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
            {
                const g6args_t q = late(G6_ARGS);
                if (q->p.bias) bias4 = load_bias4(q->p.bias, 0, q->p.N, c_n0 + wn0);
            }
            stage(X0());
            // tile finished.  Its last stage's slot (`buf` was advanced past it) is free: every wave read it before the last
            // barrier; the three younger slots hold / receive the next tile's first stages
            const int freed = slot_of_new(buf);
            char* img = wave < g6::IMG_IN_SLOT ? smem + freed * STAGE + wave * g6::IMG
                                               : smem + T::SMEM + (wave - g6::IMG_IN_SLOT) * g6::IMG;
            char* bstrip = wave < 4 ? smem + freed * STAGE + g6::IMG_IN_SLOT * g6::IMG + wave * g6::BSTRIP
                                    : smem + T::SMEM + 2 * g6::IMG + (wave - 4) * g6::BSTRIP;
            // dynamic order: the index of the tile AFTER the one already being loaded is requested here and read behind the
            // epilogue (every wave read the previous hand-over word three barriers ago)
            const bool ask = dynamic && wave == 0 && l_valid && lane_now() == 0;
            if (ask) ticket = atomicAdd(late(G6_ARGS)->ctr + xcd, 1u);
            if (sk && c_kbeg > 0) {
                // stream-K, a piece that does not start its tile: the accumulators go to this range's slab (write-through
                // stores, every wave waits for its own), then one lane raises the range's flag.  No epilogue: the workgroup
                // that owns the tile's first stages adds the slab to its own sums.  (vmcnt(0) also retires the ring's loads:
                // the relaxed waits of the next two stages are trivially met.)
                store_partial();
                wait_vmcnt_le<0>();
                raw_barrier();
                raise_flag();
            } else {
                if (sk && c_cnt < nk) {
                    // stream-K, the piece that STARTS a cut tile (always the last piece of a range): the tile's later stages
                    // were multiplied elsewhere.  Their sums are added, and the epilogue run, BEHIND the loop: with the
                    // gathering loop in here hipcc's register allocation of the whole k-loop collapsed (136 dwords of scratch)
                    // (nothing a vector-memory load produced may live out of the loop - bias4 included: it would be a pending
                    // load at the loop header, i.e. an s_waitcnt vmcnt(0) in every k-step)
                    finish = true;
                    break;
                }
                tile_epilogue<EPI, RES, TM>(epi_args(), acc, img, bstrip, bias4, c_m0 + wm0, c_n0 + wn0);
            }
            if (!l_valid) break;            // the load side ran off the list: nothing real is in flight
            if (ask) lds_st32(handover, (uint32_t)nslots + ticket);
            c_m0 = l_m0;
            c_n0 = l_n0;
            c_kbeg = 0;
            c_cnt = n_cnt;
            zero_acc();
            {
                const int ln = lane_now();
                l31 = ln & 31;
                lhi = ln >> 5;
            }
            setup_offsets();
            raw_barrier();                  // the images are dead: stage + 4's loads may overwrite the freed slot
            load_frags(buf, 0, fa[0], fw[0]);
            // the first two stages of the tile: the epilogue's stores sit between the ring loads in the vmcnt queue
            stage(XS());
            stage(XS());
            k = 2;
        }
        if (sk && finish) {
            const int freed = slot_of_new(buf);
            char* img = wave < g6::IMG_IN_SLOT ? smem + freed * STAGE + wave * g6::IMG
                                               : smem + T::SMEM + (wave - g6::IMG_IN_SLOT) * g6::IMG;
            char* bstrip = wave < 4 ? smem + freed * STAGE + g6::IMG_IN_SLOT * g6::IMG + wave * g6::BSTRIP
                                    : smem + T::SMEM + 2 * g6::IMG + (wave - 4) * g6::BSTRIP;
            gather_partials(c_cnt);
            f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
            const g6args_t q = late(G6_ARGS);
            if (q->p.bias) bias4 = load_bias4(q->p.bias, 0, q->p.N, c_n0 + wn0);
            tile_epilogue<EPI, RES, TM>(epi_args(), acc, img, bstrip, bias4, c_m0 + wm0, c_n0 + wn0);
        }
    };
    if (grpA) run(std::true_type()); else run(std::false_type());
    wait_vmcnt_le<0>();   // the zero-fill loads issued past the end of the list still write this workgroup's LDS
    leave();
}

// mode: 0 static, 1 dynamic (ctr), 2 stream-K (ctr + slabs)
template <int EPI, int RES, int VAR, int SK>
static int launch6(const GemmParams& p, uint32_t bA, uint32_t bW, int mode, uint32_t* ctr, float* slabs, int max_wg,
                   hipStream_t stream, int col_outer) {
    constexpr int BM = 256, BN = g5::BN;
    const int tM = (p.M + BM - 1) / BM, tN = (p.N + BN - 1) / BN, nk = p.K / g5::BKT;
    allow_big_smem(gemm6_kernel<EPI, RES, VAR, SK>, g6::SMEM);
    // super-tile of the XCD-local order: as launch5 (sm x sn ~ the 32 workgroups of an XCD, least operand rows per tile)
    const int rows_per_xcd = (tM + 7) / 8;
    int flat = tM < 8;
    int sm = 1, sn = 1;
    const bool w_resident = (size_t)p.N * p.K * 2 <= (size_t)3 << 20;
    if (!w_resident && !flat) {
        long best = -1;
        for (int c = 1; c <= tN && c <= 16; ++c) {
            if (tN % c) continue;
            int r = std::max(1, std::min(rows_per_xcd, (32 + c / 2) / c));
            long cost = ((long)r * BM + (long)c * BN) * 1000 / ((long)r * c);
            if (best < 0 || cost < best) best = cost, sm = r, sn = c;
        }
    }
    if (col_outer && !flat && sn < tN && tN % sn == 0) flat = 2;     // (tile_of: column blocks outermost)
    // workgroups per XCD: one per CU, no more than the longest per-XCD list (stream-K: than its stages / 2 PMIN)
    int longest = 0, shortest = 1 << 30;
    for (int x = 0; x < 8; ++x) {
        const int len = flat == 1 ? (tM * tN - x + 7) / 8 : ((tM - x + 7) / 8) * tN;
        longest = std::max(longest, len);
        if (len > 0) shortest = std::min(shortest, len);
    }
    int nslots = std::min(32, longest);
    if (mode == 2) nslots = std::max(1, std::min(32, shortest * nk / (2 * g6::PMIN)));
    if (max_wg > 0) nslots = std::max(1, std::min(nslots, max_wg / 8));
    const size_t outc = p.epi ? (size_t)p.N / 2 : (size_t)p.N;
    const size_t bytesC = ((size_t)(p.M - 1) * p.ldc + outc) * 2, bytesR = p.R ? ((size_t)(p.M - 1) * p.ldr + outc) * 2 : 0;
    if (bytesC > 0x7FFFFFF0u || bytesR > 0x7FFFFFF0u) return MC_ERR_UNSUPPORTED;
    G6Args a;
    a.p = p;
    a.bytesA = bA; a.bytesW = bW; a.bytesC = (uint32_t)bytesC; a.bytesR = (uint32_t)bytesR;
    a.bytesB = p.bias ? (uint32_t)p.N * 4u : 0u;
    a.tilesM = tM; a.tilesN = tN; a.sm = sm; a.sn = sn;
    a.mode = mode; a.flat = flat;
    a.ctr = ctr; a.slabs = slabs;
    a.bytesS = mode == 2 ? (uint32_t)(8u * 32u * g6::SLAB_FLOATS * 4u) : 0u;
    MC_LAUNCH((gemm6_kernel<EPI, RES, VAR, SK>), dim3((unsigned)(8 * nslots)), dim3(512), g6::SMEM, stream, a);
    return MC_LAST_ERROR() ? MC_ERR_LAUNCH : MC_OK;
}

size_t gemm6_slab_bytes() { return (size_t)8 * 32 * g6::SLAB_FLOATS * 4; }
size_t gemm6_counter_bytes() { return (size_t)g6::CTR_WORDS * 4; }

// Persistent tile loop for DENSE problems (one activation source, one bias row).  var bit 0: drain the epilogue's stores at
// the first k-step (no in-order assumption between stores and loads).  mode 0: static tile order; 1: dynamic (ctr: zeroed
// counter block); 2: stream-K (ctr + slabs: the stages of every XCD's tile list are dealt evenly, tiles are cut along k where
// a range ends, partial sums meet in `slabs`).  max_wg: cap on the grid (0 = one workgroup per CU).  MC_ERR_UNSUPPORTED for what
// stays on gemm5 / gemm3.
int gemm6_dispatch(const GemmParams& p, int var, int mode, uint32_t* ctr, float* slabs, int max_wg, hipStream_t stream) {
    const size_t bytesA = ((size_t)p.M * (size_t)p.lda) * 2;
    const size_t bytesW = (size_t)p.N * p.K * 2;
    const size_t lim = 0x7FFFFFF0u;
    if (bytesA > lim || bytesW > lim) return MC_ERR_UNSUPPORTED;
    if (p.A2 || p.c1 != p.K) return MC_ERR_UNSUPPORTED;
    if (p.bias && p.rows_per_batch < p.M) return MC_ERR_UNSUPPORTED;
    if (p.ws || p.splits > 1) return MC_ERR_UNSUPPORTED;
    if ((p.N & 7) || (p.ldc & 7) || (p.R && (p.ldr & 7)) || p.K % g5::BKT) return MC_ERR_UNSUPPORTED;
    if (p.K / g5::BKT < g6::NK_MIN) return MC_ERR_UNSUPPORTED;
    if (p.epi == 1 && ((p.N & 15) || p.R)) return MC_ERR_UNSUPPORTED;
    if (mode != 2 && (p.M + 255) / 256 < 8) return MC_ERR_UNSUPPORTED;     // every XCD owns at least one row of tiles
    if ((mode == 1 && !ctr) || (mode == 2 && (!ctr || !slabs)) || mode < 0 || mode > 2) return MC_ERR_SHAPE;
#define MC_G6A(E, R, V, S) launch6<E, R, V, S>(p, (uint32_t)bytesA, (uint32_t)bytesW, mode, ctr, slabs, max_wg, stream, (var & 4) != 0)
#define MC_G6(E, R) (mode == 2 ? MC_G6A(E, R, 0, 1) : (var & 1) ? MC_G6A(E, R, 1, 0) : MC_G6A(E, R, 0, 0))
    if (p.epi == 1) return MC_G6(1, 0);
    if (p.R) return MC_G6(0, 1);
    return MC_G6(0, 0);
#undef MC_G6
#undef MC_G6A
}

}  // namespace mc
