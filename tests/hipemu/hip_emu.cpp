// hip_emu.cpp -- fiber scheduler behind hip_emu.h.  TEST INFRASTRUCTURE ONLY (see hip_emu.h).
#include "hip_emu.h"

#include <sys/mman.h>

#include <condition_variable>

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
    f.coll = 0;
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

// persistent worker pool: launches are frequent and small, thread + fiber-stack creation is not
namespace {
struct Job {
    dim3 grid, block;
    size_t smem = 0;
    const std::function<void()>* body = nullptr;
    std::atomic<long> next{0};
    long nblocks = 0;
};

void run_job_on_this_thread(Job& job) {
    static thread_local Block blk;       // fibers/stacks cached per OS thread
    Block& b = blk;
    const int nthr = int(job.block.x * job.block.y * job.block.z);
    if (int(b.fibers.size()) < nthr) b.fibers.resize(nthr);
    b.n = nthr;
    b.waves.resize((nthr + kWave - 1) / kWave);
    b.bdim = job.block;
    b.gdim = job.grid;
    b.dyn_smem.resize(job.smem + 16);
    b.body = job.body;
    for (int t = 0; t < nthr; ++t) {
        Fiber& f = b.fibers[t];
        f.linear = t;
        f.tid = dim3(t % job.block.x, (t / job.block.x) % job.block.y, t / (job.block.x * job.block.y));
    }
    for (;;) {
        long i = job.next.fetch_add(1);
        if (i >= job.nblocks) break;
        dim3 bid(unsigned(i % job.grid.x), unsigned((i / job.grid.x) % job.grid.y),
                 unsigned(i / (long(job.grid.x) * job.grid.y)));
        run_block(b, bid);
    }
}

struct Pool {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    Job* job = nullptr;
    unsigned long epoch = 0;
    int wanted = 0, running = 0;
    bool stop = false;

    explicit Pool(int n) {
        for (int i = 0; i < n; ++i) threads.emplace_back([this, i] { loop(i); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> l(mu); stop = true; }
        cv_work.notify_all();
        for (auto& t : threads) t.join();
    }
    void loop(int idx) {
        unsigned long seen = 0;
        for (;;) {
            Job* j;
            {
                std::unique_lock<std::mutex> l(mu);
                cv_work.wait(l, [&] { return stop || (epoch != seen && idx < wanted); });
                if (stop) return;
                seen = epoch;
                j = job;
            }
            run_job_on_this_thread(*j);
            {
                std::lock_guard<std::mutex> l(mu);
                if (--running == 0) cv_done.notify_all();
            }
        }
    }
};
}  // namespace

void launch_impl(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
    const int nthr = int(block.x * block.y * block.z);
    const long nblocks = long(grid.x) * grid.y * grid.z;
    if (nthr <= 0 || nblocks <= 0) return;
    int hw = int(std::max(1u, std::thread::hardware_concurrency()));
    if (const char* e = getenv("HIPEMU_THREADS")) hw = std::max(1, atoi(e));
    static Pool pool(hw > 1 ? hw - 1 : 0);              // the calling thread works too
    Job job;
    job.grid = grid; job.block = block; job.smem = smem; job.body = &body; job.nblocks = nblocks;
    const int helpers = int(std::min<long>(nblocks - 1, (long)pool.threads.size()));
    if (helpers <= 0) {
        run_job_on_this_thread(job);
        return;
    }
    // helpers pull blocks from the shared counter while the caller does the same
    {
        std::unique_lock<std::mutex> l(pool.mu);
        pool.job = &job;
        pool.wanted = helpers;
        pool.running = helpers;
        ++pool.epoch;
    }
    pool.cv_work.notify_all();
    run_job_on_this_thread(job);
    {
        std::unique_lock<std::mutex> l(pool.mu);
        pool.cv_done.wait(l, [&] { return pool.running == 0; });
        pool.wanted = 0;
    }
}

}  // namespace hipemu
