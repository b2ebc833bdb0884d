// hip_emu.h -- a minimal "HIP on host fibers" shim.  TEST INFRASTRUCTURE ONLY.
//
// There is no GPU in the development container, and GPU minutes are scarce.  This header lets the
// SAME kernel sources that hipcc compiles for gfx950 (densereg_amd/csrc/*.hip) be compiled with a
// host clang++ (-DDR_EMU) and executed thread-by-thread on CPU fibers, so that index math, LDS
// layouts, MFMA fragment maps and the C-ABI plumbing can be unit-tested without a device:
//   * every GPU thread of a block is a fiber (own stack) on ONE OS thread; blocks are spread over
//     OS threads; __syncthreads / wave shuffles / MFMA are rendezvous points between fibers;
//   * MFMA builtins follow the gfx950 lane->element maps of cdna_hip_programming.md section 3
//     (A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D col=l&31,row=(r&3)+8*(r>>2)+4*(l>>5) for 32x32x2);
//   * nothing here is faster than anything: it exists to find bugs, not to run the product.
// The product (densereg_amd/) never loads a library built from this header; the loader refuses
// anything but libdensereg_hip.so.  Timing, memory-model and codegen questions need a real MI355X.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };

namespace hipemu {

constexpr int kWave = 64;
constexpr size_t kStackBytes = 96 * 1024;

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = false;
    dim3 tid;
    int linear = 0;
    unsigned coll = 0;      // number of wave collectives this lane has executed (selects the slot bank)
};

struct Wave {
    int count = 0;
    int gen = 0;
    int alive = 0;
    // two slot banks, alternated per collective: a lane can be at most one collective ahead of the
    // slowest lane of its wave, so one rendezvous per collective is enough (write bank p, sync, read).
    alignas(16) uint32_t slot[2][kWave][8];
};

struct Block {
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    int n = 0, cur = 0, alive = 0;
    int bar_count = 0, bar_gen = 0;
    void* main_sp = nullptr;
    dim3 bid, bdim, gdim;
    std::vector<char> dyn_smem;
    const std::function<void()>* body = nullptr;
    bool failed = false;
    ~Block();
};

extern thread_local Block* g_blk;
extern "C" void hipemu_switch(void** save_sp, void* load_sp);

inline Fiber& cur_fiber() { return g_blk->fibers[g_blk->cur]; }

// switch from the current fiber to the next runnable one (round robin)
inline void yield() {
    Block* b = g_blk;
    int me = b->cur;
    int nx = me;
    for (int i = 0; i < b->n; ++i) {
        nx = (nx + 1 == b->n) ? 0 : nx + 1;
        if (!b->fibers[nx].done) break;
    }
    if (nx == me) return;
    b->cur = nx;
    hipemu_switch(&b->fibers[me].sp, b->fibers[nx].sp);
}

inline void sync_block() {
    Block* b = g_blk;
    int gen = b->bar_gen;
    if (++b->bar_count >= b->alive) {
        b->bar_count = 0;
        b->bar_gen++;
        return;
    }
    long spins = 0;
    while (b->bar_gen == gen) {
        yield();
        if (++spins > 100000000L) { fprintf(stderr, "hipemu: __syncthreads deadlock\n"); abort(); }
    }
}

inline Wave& my_wave() { return g_blk->waves[cur_fiber().linear / kWave]; }
inline int lane_id() { return cur_fiber().linear % kWave; }

inline void sync_wave() {
    Wave& w = my_wave();
    int gen = w.gen;
    if (++w.count >= w.alive) {
        w.count = 0;
        w.gen++;
        return;
    }
    long spins = 0;
    while (w.gen == gen) {
        yield();
        if (++spins > 100000000L) { fprintf(stderr, "hipemu: divergent wave collective (deadlock)\n"); abort(); }
    }
}

void fiber_exit();   // marks done, releases barriers, switches away (never returns)
void launch_impl(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);

template <class T>
inline T exchange(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "exchange width");
    Wave& w = my_wave();
    int l = lane_id();
    const unsigned bank = (cur_fiber().coll++) & 1u;
    std::memcpy(w.slot[bank][l], &v, sizeof(T));
    sync_wave();
    T r;
    std::memcpy(&r, w.slot[bank][src_lane & (kWave - 1)], sizeof(T));
    return r;
}

}  // namespace hipemu

// ---- built-in variables --------------------------------------------------------------------
#define threadIdx (hipemu::cur_fiber().tid)
#define blockIdx (hipemu::g_blk->bid)
#define blockDim (hipemu::g_blk->bdim)
#define gridDim (hipemu::g_blk->gdim)
#define warpSize 64

static inline void __syncthreads() { hipemu::sync_block(); }
static inline void __builtin_amdgcn_s_barrier() { hipemu::sync_block(); }
static inline void __builtin_amdgcn_wave_barrier() { hipemu::sync_wave(); }
static inline void __threadfence() {}
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __builtin_amdgcn_sched_barrier(int) {}

template <class T> static inline T __shfl(T v, int src, int = 64) { return hipemu::exchange(v, src); }
template <class T> static inline T __shfl_xor(T v, int m, int = 64) { return hipemu::exchange(v, hipemu::lane_id() ^ m); }
template <class T> static inline T __shfl_down(T v, int d, int = 64) {
    int s = hipemu::lane_id() + d;
    return hipemu::exchange(v, s < 64 ? s : hipemu::lane_id());
}
template <class T> static inline T __shfl_up(T v, int d, int = 64) {
    int s = hipemu::lane_id() - d;
    return hipemu::exchange(v, s >= 0 ? s : hipemu::lane_id());
}
static inline unsigned long long __ballot(int pred) {
    unsigned long long m = 0;
    for (int i = 0; i < 64; i += 1) {
        // one exchange per lane is slow but simple; ballots are rare in this code base
        int p = hipemu::exchange(pred ? 1 : 0, i);
        if (p) m |= (1ull << i);
    }
    return m;
}
static inline int __all(int pred) { return __ballot(pred) == ~0ull; }
static inline int __any(int pred) { return __ballot(pred) != 0ull; }
static inline int __builtin_amdgcn_readfirstlane(int v) { return hipemu::exchange(v, 0); }

// ---- atomics (blocks run on different OS threads) -------------------------------------------
static inline float atomicAdd(float* p, float v) {
    auto* a = reinterpret_cast<std::atomic<uint32_t>*>(p);
    uint32_t old = a->load(std::memory_order_relaxed), neu;
    float f;
    do { std::memcpy(&f, &old, 4); f += v; std::memcpy(&neu, &f, 4); } while (!a->compare_exchange_weak(old, neu));
    std::memcpy(&f, &old, 4);
    return f;
}
static inline double atomicAdd(double* p, double v) {
    auto* a = reinterpret_cast<std::atomic<uint64_t>*>(p);
    uint64_t old = a->load(std::memory_order_relaxed), neu;
    double f;
    do { std::memcpy(&f, &old, 8); f += v; std::memcpy(&neu, &f, 8); } while (!a->compare_exchange_weak(old, neu));
    std::memcpy(&f, &old, 8);
    return f;
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}

// ---- MFMA ------------------------------------------------------------------------------------
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: D(32x32) = A(32x2) B(2x32) + C, k-ordered fma chain (bit-exact on HW)
static inline hipemu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, hipemu_f32x16 c, int, int, int) {
    hipemu::Wave& w = hipemu::my_wave();
    int l = hipemu::lane_id();
    const unsigned bank = (hipemu::cur_fiber().coll++) & 1u;
    std::memcpy(&w.slot[bank][l][0], &a, 4);
    std::memcpy(&w.slot[bank][l][1], &b, 4);
    hipemu::sync_wave();
    hipemu_f32x16 d = c;
    int j = l & 31;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av, bv;
            std::memcpy(&av, &w.slot[bank][i + 32 * k][0], 4);
            std::memcpy(&bv, &w.slot[bank][j + 32 * k][1], 4);
            acc = std::fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    return d;
}

// v_mfma_f32_32x32x16_bf16 (gfx950): lane l holds 8 bf16 of row l&31, k = 8*(l>>5) + 0..7, for both operands; D as above.
// Products of two bf16 are exact in fp32; the accumulation order inside the instruction is not architected (k order here).
typedef __bf16 hipemu_bf16x8 __attribute__((ext_vector_type(8)));
static inline hipemu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x16 c, int, int, int) {
    hipemu::Wave& w = hipemu::my_wave();
    int l = hipemu::lane_id();
    const unsigned bank = (hipemu::cur_fiber().coll++) & 1u;
    std::memcpy(&w.slot[bank][l][0], &a, 16);
    std::memcpy(&w.slot[bank][l][4], &b, 16);
    hipemu::sync_wave();
    auto bf = [](uint32_t word, int half) { uint32_t u = (half ? (word >> 16) : (word & 0xFFFFu)) << 16; float f; std::memcpy(&f, &u, 4); return f; };
    hipemu_f32x16 d = c;
    int j = l & 31;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            const int src = 32 * (k >> 3), e = k & 7;
            acc = std::fmaf(bf(w.slot[bank][i + src][e >> 1], e & 1), bf(w.slot[bank][j + src][4 + (e >> 1)], e & 1), acc);
        }
        d[r] = acc;
    }
    return d;
}

// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D col=l&15,row=(l>>4)*4+r
static inline hipemu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
    hipemu::Wave& w = hipemu::my_wave();
    int l = hipemu::lane_id();
    const unsigned bank = (hipemu::cur_fiber().coll++) & 1u;
    std::memcpy(&w.slot[bank][l][0], &a, 4);
    std::memcpy(&w.slot[bank][l][1], &b, 4);
    hipemu::sync_wave();
    hipemu_f32x4 d = c;
    int j = l & 15;
    for (int r = 0; r < 4; ++r) {
        int i = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            float av, bv;
            std::memcpy(&av, &w.slot[bank][i + 16 * k][0], 4);
            std::memcpy(&bv, &w.slot[bank][j + 16 * k][1], 4);
            acc = std::fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    return d;
}

// ---- runtime -----------------------------------------------------------------------------------
template <class K, class... Args>
static inline void hipemu_launch(K kernel, dim3 grid, dim3 block, size_t smem, hipStream_t, Args... args) {
    std::function<void()> body = [=]() { kernel(args...); };
    hipemu::launch_impl(grid, block, smem, body);
}

static inline char* hipemu_dyn_smem() { return hipemu::g_blk->dyn_smem.data(); }
