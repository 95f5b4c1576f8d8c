// TEST INFRASTRUCTURE ONLY — see hip_emu.h.  Fiber scheduler for one workgroup.
#include "hip_emu.h"

// Minimal x86-64 SysV context switch: callee-saved registers + stack pointer.
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch, .-hipemu_switch
)");

namespace hipemu {

thread_local BlockCtx* g_blk = nullptr;

static void fiber_entry() {
    BlockCtx* b = g_blk;
    (*b->body)();
    b = g_blk;
    Fiber& f = b->fibers[b->cur];
    f.done = true;
    b->live--;
    for (;;) hipemu_switch(&f.sp, b->sched_sp);  // never resumed
}

void run_block(BlockCtx& ctx, char* stacks) {
    g_blk = &ctx;
    ctx.live = ctx.nfib;
    ctx.bar_arrived = 0;
    ctx.bar_gen = 0;
    for (int i = 0; i < ctx.nfib; ++i) {
        Fiber& f = ctx.fibers[i];
        f.done = false;
        f.ops = 0;
        unsigned lin = (unsigned)i;
        f.tid = dim3(lin % ctx.bdim.x, (lin / ctx.bdim.x) % ctx.bdim.y, lin / (ctx.bdim.x * ctx.bdim.y));
        uintptr_t top = ((uintptr_t)(stacks + (size_t)(i + 1) * kStackBytes)) & ~(uintptr_t)15;
        top -= 8;                                   // so that rsp % 16 == 8 at fiber_entry
        void** sp = (void**)(top - 8);
        *sp = (void*)&fiber_entry;                  // return address consumed by `ret`
        sp -= 6;                                    // r15 r14 r13 r12 rbx rbp
        for (int k = 0; k < 6; ++k) sp[k] = nullptr;
        f.sp = (void*)sp;
    }
    long idle_rounds = 0;
    while (ctx.live > 0) {
        unsigned gen_before = ctx.bar_gen;
        int live_before = ctx.live;
        for (int i = 0; i < ctx.nfib; ++i) {
            if (ctx.fibers[i].done) continue;
            ctx.cur = i;
            hipemu_switch(&ctx.sched_sp, ctx.fibers[i].sp);
        }
        // a stuck barrier (divergent __syncthreads) would spin forever: bound it
        if (ctx.bar_gen == gen_before && ctx.live == live_before && ctx.bar_arrived > 0) {
            if (++idle_rounds > 1000000) {
                std::fprintf(stderr, "hipemu: workgroup (%u,%u,%u) deadlocked at a barrier\n",
                             ctx.bid.x, ctx.bid.y, ctx.bid.z);
                std::abort();
            }
        } else {
            idle_rounds = 0;
        }
    }
    g_blk = nullptr;
}

}  // namespace hipemu
