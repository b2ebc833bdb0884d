// hip_emu.cpp -- fiber scheduler behind hip_emu.h.  TEST INFRASTRUCTURE ONLY (see hip_emu.h).
#include "hip_emu.h"

#include <sys/mman.h>

#if !defined(__x86_64__)
#error "hipemu context switch is written for x86-64"
#endif

// void hipemu_switch(void** save_sp, void* load_sp): save callee-saved registers on the current
// stack, publish its sp, adopt the other stack, restore, return into the other fiber.
__asm__(
    ".text\n"
    ".globl hipemu_switch\n"
    ".type hipemu_switch,@function\n"
    "hipemu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size hipemu_switch,.-hipemu_switch\n");

namespace hipemu {

thread_local Block* g_blk = nullptr;

Block::~Block() {
    for (auto& f : fibers)
        if (f.stack) munmap(f.stack, kStackBytes);
}

static void trampoline() {
    Block* b = g_blk;
    (*b->body)();
    fiber_exit();
}

void fiber_exit() {
    Block* b = g_blk;
    Fiber& f = b->fibers[b->cur];
    f.done = true;
    b->alive--;
    Wave& w = b->waves[f.linear / kWave];
    w.alive--;
    // threads that exit no longer take part in barriers (hardware semantics)
    if (b->alive > 0 && b->bar_count >= b->alive) { b->bar_count = 0; b->bar_gen++; }
    if (w.alive > 0 && w.count >= w.alive) { w.count = 0; w.gen++; }
    if (b->alive == 0) {
        void* dummy;
        hipemu_switch(&dummy, b->main_sp);
    }
    int me = b->cur, nx = me;
    for (int i = 0; i < b->n; ++i) {
        nx = (nx + 1 == b->n) ? 0 : nx + 1;
        if (!b->fibers[nx].done) break;
    }
    b->cur = nx;
    void* dummy;
    hipemu_switch(&dummy, b->fibers[nx].sp);
    abort();
}

static void prepare_fiber(Fiber& f) {
    if (!f.stack) {
        void* p = mmap(nullptr, kStackBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) { perror("hipemu mmap"); abort(); }
        f.stack = static_cast<char*>(p);
    }
    uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + kStackBytes) & ~uintptr_t(15);
    void** s = reinterpret_cast<void**>(top);
    s[-1] = nullptr;                                  // fake return address of trampoline
    s[-2] = reinterpret_cast<void*>(&trampoline);     // 'ret' target
    for (int i = 3; i <= 8; ++i) s[-i] = nullptr;     // rbp rbx r12 r13 r14 r15
    f.sp = &s[-8];
    f.done = false;
}

static void run_block(Block& b, dim3 bid) {
    b.bid = bid;
    b.cur = 0;
    b.alive = b.n;
    b.bar_count = 0;
    b.bar_gen = 0;
    for (auto& w : b.waves) { w.count = 0; w.gen = 0; w.alive = 0; }
    for (int t = 0; t < b.n; ++t) {
        prepare_fiber(b.fibers[t]);
        b.waves[t / kWave].alive++;
    }
    g_blk = &b;
    hipemu_switch(&b.main_sp, b.fibers[0].sp);
    g_blk = nullptr;
}

void launch_impl(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
    const int nthr = int(block.x * block.y * block.z);
    const long nblocks = long(grid.x) * grid.y * grid.z;
    if (nthr <= 0 || nblocks <= 0) return;
    int nworkers = int(std::min<long>(nblocks, std::max(1u, std::thread::hardware_concurrency())));
    if (const char* e = getenv("HIPEMU_THREADS")) nworkers = std::max(1, std::min(nworkers, atoi(e)));
    std::atomic<long> next{0};
    auto worker = [&]() {
        static thread_local Block blk;       // fibers/stacks cached per OS thread
        Block& b = blk;
        if (int(b.fibers.size()) < nthr) b.fibers.resize(nthr);
        b.n = nthr;
        b.waves.resize((nthr + kWave - 1) / kWave);
        b.bdim = block;
        b.gdim = grid;
        b.dyn_smem.resize(smem + 16);
        b.body = &body;
        for (int t = 0; t < nthr; ++t) {
            Fiber& f = b.fibers[t];
            f.linear = t;
            f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
        }
        for (;;) {
            long i = next.fetch_add(1);
            if (i >= nblocks) break;
            dim3 bid(unsigned(i % grid.x), unsigned((i / grid.x) % grid.y), unsigned(i / (long(grid.x) * grid.y)));
            run_block(b, bid);
        }
    };
    if (nworkers == 1) {
        worker();
    } else {
        std::vector<std::thread> ts;
        for (int i = 0; i < nworkers; ++i) ts.emplace_back(worker);
        for (auto& t : ts) t.join();
    }
}

}  // namespace hipemu
