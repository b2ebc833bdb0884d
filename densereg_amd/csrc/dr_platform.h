// dr_platform.h -- the one place that knows whether we compile for gfx950 (hipcc, the product)
// or for the host-fiber emulator used by the CPU-side unit tests (tests/hipemu, -DDR_EMU).
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#if defined(DR_EMU)
// ---------------------------------------------------------------------------------------------
// Test build: same kernel sources, executed by tests/hipemu/hip_emu.h.  Never shipped.
// ---------------------------------------------------------------------------------------------
#include "hip_emu.h"
#define DR_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipemu_launch(kernel, grid, block, smem, stream, __VA_ARGS__)
#define DR_DYN_SMEM(name) char* name = hipemu_dyn_smem()
namespace dr { namespace rt {
inline const char* backend_name() { return "hipemu"; }
inline int set_device(int) { return 0; }
inline int device_count() { return 1; }
inline void* dmalloc(size_t n) { void* p = nullptr; if (posix_memalign(&p, 256, n ? n : 256)) return nullptr; return p; }
inline void dfree(void* p) { free(p); }
inline int memset_async(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
inline int h2d(void* d, const void* h, size_t n, hipStream_t) { memcpy(d, h, n); return 0; }
inline int d2h(void* h, const void* d, size_t n, hipStream_t) { memcpy(h, d, n); return 0; }
inline int d2d(void* d, const void* s, size_t n, hipStream_t) { memcpy(d, s, n); return 0; }
inline int sync_stream(hipStream_t) { return 0; }
inline int sync_stream_bounded(hipStream_t) { return 0; }
inline int last_error(std::string*) { return 0; }
struct Event { int dummy; };
inline Event event_create() { return Event{0}; }
inline void event_destroy(Event) {}
inline void event_record(Event, hipStream_t) {}
inline float event_elapsed_ms(Event, Event) { return 0.f; }
inline Event event_create_sync() { return Event{0}; }
inline hipStream_t stream_create() { return (hipStream_t) nullptr; }         // the emulator runs every launch synchronously
inline void stream_destroy(hipStream_t) {}
inline hipStream_t stream_create_low_priority() { return (hipStream_t) nullptr; }
inline hipStream_t stream_create_high_priority() { return (hipStream_t) nullptr; }
inline void stream_wait_event(hipStream_t, Event) {}
struct Graph { int dummy; };                                                  // no graphs on the emulator
inline bool capture_begin(hipStream_t) { return false; }
inline bool capture_end(hipStream_t, Graph*) { return false; }
inline bool graph_launch(Graph, hipStream_t) { return false; }
inline void graph_destroy(Graph) {}
inline bool allow_dyn_lds(const void*, size_t) { return true; }
}}  // namespace dr::rt
#else
// ---------------------------------------------------------------------------------------------
// Product build: HIP for gfx950.
// ---------------------------------------------------------------------------------------------
#include <hip/hip_runtime.h>

#include <chrono>
#include <thread>
#define DR_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__)
#define DR_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
namespace dr { namespace rt {
inline const char* backend_name() { return "hip-gfx950"; }
inline int set_device(int d) { return hipSetDevice(d) == hipSuccess ? 0 : -1; }
inline int device_count() { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
inline void* dmalloc(size_t n) { void* p = nullptr; if (hipMalloc(&p, n ? n : 256) != hipSuccess) return nullptr; return p; }
inline void dfree(void* p) { (void)hipFree(p); }
inline int memset_async(void* p, int v, size_t n, hipStream_t s) { return hipMemsetAsync(p, v, n, s) == hipSuccess ? 0 : -1; }
inline int h2d(void* d, const void* h, size_t n, hipStream_t s) { return hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s) == hipSuccess ? 0 : -1; }
inline int d2h(void* h, const void* d, size_t n, hipStream_t s) { return hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s) == hipSuccess ? 0 : -1; }
inline int d2d(void* d, const void* s_, size_t n, hipStream_t s) { return hipMemcpyAsync(d, s_, n, hipMemcpyDeviceToDevice, s) == hipSuccess ? 0 : -1; }
// Host-blocking wait, BOUNDED: a stream that does not drain within DR_SYNC_TIMEOUT_S seconds (default 120; 0 = wait forever) is
// reported (-2, one line on stderr) instead of hanging the caller.  ONLY for callers that act on the result (pipeline_drain,
// dr_finalize_params, the stream-destroy paths: a stream that does not drain is leaked, not destroyed under its work) -- a caller
// that goes on to read a host buffer, reuse a host table or free device memory must use sync_stream below, which never returns
// before the stream has drained.  The poll is hipStreamQuery with a short sleep: these are rare paths.
inline int sync_stream_bounded(hipStream_t s) {
    static const double limit = [] { const char* e = getenv("DR_SYNC_TIMEOUT_S"); return e ? atof(e) : 120.0; }();
    if (!(limit > 0.0) || hipPeekAtLastError() != hipSuccess) return hipStreamSynchronize(s) == hipSuccess ? 0 : -1;
    hipError_t e = hipStreamQuery(s);
    if (e == hipSuccess) return 0;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned nap_us = 5;
    while (e == hipErrorNotReady) {
        std::this_thread::sleep_for(std::chrono::microseconds(nap_us));
        if (nap_us < 200) nap_us *= 2;
        e = hipStreamQuery(s);
        if (e == hipErrorNotReady && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
            fprintf(stderr, "densereg: a stream did not drain within %.0f s (DR_SYNC_TIMEOUT_S): giving up the wait\n", limit);
            if (hipPeekAtLastError() == hipErrorNotReady) (void)hipGetLastError();
            return -2;
        }
    }
    if (hipPeekAtLastError() == hipErrorNotReady) (void)hipGetLastError();     // (a poll's "not ready" is not an error of the caller's)
    return e == hipSuccess ? 0 : -1;
}
// Host-blocking wait that NEVER returns early: what every caller that depends on completion uses (device-to-host reads, host tables
// the copy engine is still reading, dfree of buffers in flight).  A stream that exceeds DR_SYNC_TIMEOUT_S is NAMED on stderr (the
// line of sync_stream_bounded) -- and then waited for: returning there would hand the caller an unfilled buffer as success.
inline int sync_stream(hipStream_t s) {
    const int rc = sync_stream_bounded(s);
    if (rc != -2) return rc;
    return hipStreamSynchronize(s) == hipSuccess ? 0 : -1;
}
inline int last_error(std::string* msg) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    if (msg) *msg = hipGetErrorString(e);
    return -1;
}
struct Event { hipEvent_t e; };
inline Event event_create() { Event ev{}; (void)hipEventCreate(&ev.e); return ev; }
inline void event_destroy(Event ev) { (void)hipEventDestroy(ev.e); }
inline void event_record(Event ev, hipStream_t s) { (void)hipEventRecord(ev.e, s); }
inline float event_elapsed_ms(Event a, Event b) { float ms = 0.f; (void)hipEventElapsedTime(&ms, a.e, b.e); return ms; }
// ordering-only event (no timestamps) and the library-owned side streams of the executor's lanes
inline Event event_create_sync() { Event ev{}; (void)hipEventCreateWithFlags(&ev.e, hipEventDisableTiming); return ev; }
inline hipStream_t stream_create() { hipStream_t s = nullptr; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); return s; }
inline void stream_destroy(hipStream_t s) { if (s) (void)hipStreamDestroy(s); }
// lowest-priority stream: its workgroups are dispatched after those of the caller's stream when both have work queued
inline hipStream_t stream_create_low_priority() {
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    hipStream_t s = nullptr;
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least) != hipSuccess) { (void)hipGetLastError(); return stream_create(); }
    return s;
}
// highest-priority stream.  The runtime multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default), least
// used first, one pool PER PRIORITY LEVEL: two streams that must run concurrently (the micro-step slots, pipeline.inc) are safe
// from sharing a queue with each other or with the caller's normal-priority streams when they alone populate a level.
inline hipStream_t stream_create_high_priority() {
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    hipStream_t s = nullptr;
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest) != hipSuccess) { (void)hipGetLastError(); return stream_create(); }
    return s;
}
inline void stream_wait_event(hipStream_t s, Event ev) { (void)hipStreamWaitEvent(s, ev.e, 0); }
// stream capture -> executable graph (the launch-bound inference path replays one graph instead of ~150 launches)
struct Graph { hipGraphExec_t exec; };
inline bool capture_begin(hipStream_t s) { return hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess; }
inline bool capture_end(hipStream_t s, Graph* g) {
    hipGraph_t gr = nullptr;
    if (hipStreamEndCapture(s, &gr) != hipSuccess || !gr) { (void)hipGetLastError(); return false; }
    const bool ok = hipGraphInstantiate(&g->exec, gr, nullptr, nullptr, 0) == hipSuccess;
    (void)hipGraphDestroy(gr);
    if (!ok) (void)hipGetLastError();
    return ok;
}
inline bool graph_launch(Graph g, hipStream_t s) { return hipGraphLaunch(g.exec, s) == hipSuccess; }
inline void graph_destroy(Graph g) { if (g.exec) (void)hipGraphExecDestroy(g.exec); }
// a kernel that asks for more than 64 KB of dynamic LDS per workgroup (gfx950 has 160 KB per CU) must be given the attribute once
inline bool allow_dyn_lds(const void* kernel, size_t bytes) {
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
    return true;
}
}}  // namespace dr::rt
#endif

// Pin kernel arguments into scalar registers HERE, as one batch of scalar loads behind a single wait: hipcc loads struct
// arguments lazily, a field at a time behind the branches that first need it, and in the prologue of a short kernel every such
// load is another serialised trip to the scalar cache (a miss goes to L2 / HBM: the argument buffer was just written by the host).
#if defined(DR_EMU)
#define DR_PIN_ARGS(...) ((void)0)
#else
#define DR_PIN1(a) asm volatile("" ::"s"(a))
#define DR_PIN_ARGS_1(a) DR_PIN1(a)
#define DR_PIN_ARGS_2(a, ...) DR_PIN1(a); DR_PIN_ARGS_1(__VA_ARGS__)
#define DR_PIN_ARGS_3(a, ...) DR_PIN1(a); DR_PIN_ARGS_2(__VA_ARGS__)
#define DR_PIN_ARGS_4(a, ...) DR_PIN1(a); DR_PIN_ARGS_3(__VA_ARGS__)
#define DR_PIN_ARGS_5(a, ...) DR_PIN1(a); DR_PIN_ARGS_4(__VA_ARGS__)
#define DR_PIN_ARGS_6(a, ...) DR_PIN1(a); DR_PIN_ARGS_5(__VA_ARGS__)
#define DR_PIN_ARGS_7(a, ...) DR_PIN1(a); DR_PIN_ARGS_6(__VA_ARGS__)
#define DR_PIN_ARGS_8(a, ...) DR_PIN1(a); DR_PIN_ARGS_7(__VA_ARGS__)
#define DR_PIN_ARGS_9(a, ...) DR_PIN1(a); DR_PIN_ARGS_8(__VA_ARGS__)
#define DR_PIN_ARGS_10(a, ...) DR_PIN1(a); DR_PIN_ARGS_9(__VA_ARGS__)
#define DR_PIN_ARGS_11(a, ...) DR_PIN1(a); DR_PIN_ARGS_10(__VA_ARGS__)
#define DR_PIN_ARGS_12(a, ...) DR_PIN1(a); DR_PIN_ARGS_11(__VA_ARGS__)
#define DR_PIN_ARGS_13(a, ...) DR_PIN1(a); DR_PIN_ARGS_12(__VA_ARGS__)
#define DR_PIN_ARGS_14(a, ...) DR_PIN1(a); DR_PIN_ARGS_13(__VA_ARGS__)
#define DR_PIN_ARGS_15(a, ...) DR_PIN1(a); DR_PIN_ARGS_14(__VA_ARGS__)
#define DR_PIN_ARGS_16(a, ...) DR_PIN1(a); DR_PIN_ARGS_15(__VA_ARGS__)
#define DR_PIN_ARGS_17(a, ...) DR_PIN1(a); DR_PIN_ARGS_16(__VA_ARGS__)
#define DR_PIN_ARGS_18(a, ...) DR_PIN1(a); DR_PIN_ARGS_17(__VA_ARGS__)
#define DR_PIN_ARGS_19(a, ...) DR_PIN1(a); DR_PIN_ARGS_18(__VA_ARGS__)
#define DR_PIN_ARGS_20(a, ...) DR_PIN1(a); DR_PIN_ARGS_19(__VA_ARGS__)
#define DR_PIN_COUNT(_1, _2, _3, _4, _5, _6, _7, _8, _9, _10, _11, _12, _13, _14, _15, _16, _17, _18, _19, _20, N, ...) N
#define DR_PIN_CAT(a, b) a##b
#define DR_PIN_SEL(n) DR_PIN_CAT(DR_PIN_ARGS_, n)
#define DR_PIN_ARGS(...) do { DR_PIN_SEL(DR_PIN_COUNT(__VA_ARGS__, 20, 19, 18, 17, 16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1))(__VA_ARGS__); } while (0)
#endif

// 16-byte asynchronous global -> LDS copy (LDS-DMA): lane l of the wave writes lds_wave_base + 16*l; `src` is per lane.
#if defined(DR_EMU)
static inline void dr_glds16(const float* src, float* lds_wave_base) { memcpy(lds_wave_base + (threadIdx.x & 63) * 4, src, 16); }
#else
__device__ __forceinline__ void dr_glds16(const float* src, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
#endif

// Cross-workgroup hand-off inside one kernel (train_kernels.h: BatchReNorm coefficient look-back): agent-scope loads that
// bypass the non-coherent cache levels, and a polite spin.
#if defined(DR_EMU)
#include <sched.h>
static inline int dr_load_agent_i32(const int* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
static inline float dr_load_agent_f32(const float* p) { float v; __atomic_load(p, &v, __ATOMIC_ACQUIRE); return v; }
static inline void dr_spin_pause() { sched_yield(); }
static inline void dr_acquire_agent() { __atomic_thread_fence(__ATOMIC_ACQUIRE); }
#else
// acquire at agent scope: drops the non-coherent lines of this CU's L1 and this XCD's L2 (no write-back)
__device__ __forceinline__ void dr_acquire_agent() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
__device__ __forceinline__ int dr_load_agent_i32(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float dr_load_agent_f32(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void dr_spin_pause() { __builtin_amdgcn_s_sleep(2); }
#endif

// two 16-bit halves of two words in one instruction: {lo16(a), lo16(b)} (odd = 0) or {hi16(a), hi16(b)} (odd = 1), a in the
// low half of the result (v_perm_b32)
#if defined(DR_EMU)
static inline unsigned dr_pack_halves(unsigned a, unsigned b, int odd) { return odd ? ((a >> 16) | (b & 0xFFFF0000u)) : ((a & 0xFFFFu) | (b << 16)); }
#else
__device__ __forceinline__ unsigned dr_pack_halves(unsigned a, unsigned b, int odd) {
    return odd ? __builtin_amdgcn_perm(b, a, 0x07060302u) : __builtin_amdgcn_perm(b, a, 0x05040100u);
}
#endif

// LDS transpose read (gfx950 ds_read_b64_tr_b16): within each group of 16 lanes, lane i SUPPLIES the address of a 4-element
// (8-byte) chunk and lane c RECEIVES element (c & 3) of the chunks of lanes 4j + (c >> 2), j = 0..3 -- with the chunks laid
// out as a [4 k][16 columns] block (chunk i = k i/4, columns 4(i%4)..+3; any row stride) lane c gets the four k of column c:
// an MFMA operand fragment straight out of a row-major image.  Returned as two words: {e0 | e1 << 16, e2 | e3 << 16}.
#if defined(DR_EMU)
static inline uint2 dr_lds_read_tr16(const void* chunk) {
    unsigned short e[4];
    const int l = hipemu::lane_id(), gb = l & ~15, c = l & 15;
    for (int j = 0; j < 4; ++j) {
        const uintptr_t a = hipemu::exchange((uintptr_t)chunk, gb + 4 * j + (c >> 2));
        e[j] = reinterpret_cast<const unsigned short*>(a)[c & 3];
    }
    return make_uint2((unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16));
}
#else
typedef short dr_i16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 dr_lds_read_tr16(const void* chunk) {
    const dr_i16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) dr_i16x4*)chunk);
    return __builtin_bit_cast(uint2, r);
}
#endif

// 16-byte load from an address that is only 4- / 8-byte aligned (rows of bf16 tensors: the row stride is a multiple of four
// ELEMENTS): one global_load_dwordx4 on gfx950 (dword alignment is all it needs), a plain memcpy on the host emulator
__host__ __device__ static inline float4 dr_load16_a4(const void* p) { float4 v; __builtin_memcpy(&v, p, 16); return v; }

typedef float dr_f32x16 __attribute__((ext_vector_type(16)));
typedef float dr_f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 dr_bf16x8 __attribute__((ext_vector_type(8)));      // one operand of v_mfma_f32_32x32x16_bf16
typedef __bf16 dr_bf16x4 __attribute__((ext_vector_type(4)));
typedef float dr_f32x4 __attribute__((ext_vector_type(4)));
// DR_NT (experiment switch of the build, bit mask): non-temporal hints on streams that are written once and read once --
//   4 the conv epilogues' output stores, 8 the fold launch's slab loads, 16 the weight-gradient kernels' slab stores
// (the BatchReNorm passes have their own: DR_BN_NT, train_kernels.h)
#ifndef DR_NT
#define DR_NT 0
#endif
#if defined(DR_EMU)
#define DR_NT_STORE(bit, ptr, v) (*(ptr) = (v))
#define DR_NT_LOAD(bit, ptr) (*(ptr))
#else
#define DR_NT_STORE(bit, ptr, v) do { if constexpr ((DR_NT & (bit)) != 0) __builtin_nontemporal_store((v), (ptr)); else *(ptr) = (v); } while (0)
#define DR_NT_LOAD(bit, ptr) (((DR_NT & (bit)) != 0) ? __builtin_nontemporal_load(ptr) : *(ptr))
#endif

__host__ __device__ static inline int dr_ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ static inline int dr_round_up(int a, int b) { return dr_ceil_div(a, b) * b; }
