// TEST INFRASTRUCTURE ONLY: fiber scheduler behind hip_emu.h (x86-64 SysV).
#include "hip_emu.h"

#include <vector>

namespace emu {

extern "C" void emu_switch(void** save_sp, void* next_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
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
.size emu_switch,.-emu_switch
)");

constexpr size_t kStack = 96 * 1024;

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = true;
    FiberState st;
};

struct Worker {
    std::vector<Fiber> fibers;
    void* main_sp = nullptr;
    int cur = -1;
    int nthreads = 0;
    BlockState blk;
    const std::function<void()>* body = nullptr;
    // barriers
    int bar_count = 0, bar_gen = 0;
    int wbar_count[EMU_MAX_WAVES] = {}, wbar_gen[EMU_MAX_WAVES] = {};
    std::vector<float> smem;
};

static thread_local Worker* tl_worker = nullptr;

static Worker& worker()
{
    if (!tl_worker) tl_worker = new Worker();
    return *tl_worker;
}

FiberState& cur_fiber() { Worker& w = worker(); return w.fibers[w.cur].st; }
BlockState& cur_block() { return worker().blk; }

static void yield_to_main()
{
    Worker& w = worker();
    Fiber& f = w.fibers[w.cur];
    emu_switch(&f.sp, w.main_sp);
}

void block_barrier()
{
    Worker& w = worker();
    const int gen = w.bar_gen;
    if (++w.bar_count == w.nthreads) { w.bar_count = 0; w.bar_gen++; return; }
    while (w.bar_gen == gen) yield_to_main();
}

void wave_barrier()
{
    Worker& w = worker();
    const int wave = w.fibers[w.cur].st.tidx.x >> 6;
    const int nw = (w.nthreads - wave * 64) < 64 ? (w.nthreads - wave * 64) : 64;
    const int gen = w.wbar_gen[wave];
    if (++w.wbar_count[wave] == nw) { w.wbar_count[wave] = 0; w.wbar_gen[wave]++; return; }
    while (w.wbar_gen[wave] == gen) yield_to_main();
}

static void trampoline()
{
    Worker& w = worker();
    (*w.body)();
    w.fibers[w.cur].done = true;
    yield_to_main();
    std::abort();   // never resumed
}

static void run_block(Worker& w)
{
    const int n = w.nthreads;
    if ((int)w.fibers.size() < n) {
        const size_t old = w.fibers.size();
        w.fibers.resize(n);
        for (size_t i = old; i < (size_t)n; ++i) w.fibers[i].stack = (char*)std::malloc(kStack);
    }
    for (int i = 0; i < n; ++i) {
        Fiber& f = w.fibers[i];
        f.done = false;
        f.st.tidx = dim3(i, 0, 0);
        f.st.op_parity = 0;
        uintptr_t top = ((uintptr_t)(f.stack + kStack)) & ~(uintptr_t)15;
        void** sp = (void**)top;
        *(--sp) = nullptr;                 // fake return address slot (keeps entry alignment)
        *(--sp) = (void*)&trampoline;      // popped by `ret`
        for (int r = 0; r < 6; ++r) *(--sp) = nullptr;   // rbp rbx r12-r15
        f.sp = (void*)sp;
    }
    w.bar_count = 0;
    for (int k = 0; k < EMU_MAX_WAVES; ++k) w.wbar_count[k] = 0;
    int remaining = n;
    while (remaining > 0) {
        for (int i = 0; i < n; ++i) {
            if (w.fibers[i].done) continue;
            w.cur = i;
            emu_switch(&w.main_sp, w.fibers[i].sp);
            if (w.fibers[i].done) --remaining;
        }
    }
}

void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body)
{
    const long nblocks = (long)grid.x * grid.y * grid.z;
#pragma omp parallel for schedule(dynamic)
    for (long bi = 0; bi < nblocks; ++bi) {
        Worker& w = worker();
        w.nthreads = (int)block.x;
        w.body = &body;
        if (w.smem.size() * sizeof(float) < smem_bytes + 64) w.smem.resize(smem_bytes / sizeof(float) + 16);
        // 16-byte aligned dynamic LDS base
        uintptr_t base = ((uintptr_t)w.smem.data() + 15) & ~(uintptr_t)15;
        w.blk.dyn_smem = (float*)base;
        w.blk.bdim = block;
        w.blk.gdim = grid;
        w.blk.bidx = dim3((unsigned)(bi % grid.x), (unsigned)((bi / grid.x) % grid.y), (unsigned)(bi / ((long)grid.x * grid.y)));
        run_block(w);
    }
}

}  // namespace emu
