// Does a streaming pass hit the 256 MiB Infinity Cache when it reads a tensor the previous launch wrote?  Kernel W writes n bytes in
// ascending chunk order; kernel R reads them ascending (what every pass of the step does today) or descending (last written first:
// LRU-friendly when the tensor is larger than what the cache keeps).  Prints GB/s of R for both orders and several sizes.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mall_order.hip -o /tmp/mall_order && /tmp/mall_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kChunk = 64 * 1024;   // bytes per workgroup
__global__ __launch_bounds__(256) void wr(float4* p, long nchunks, int rev) {
    const long c = rev ? nchunks - 1 - blockIdx.x : blockIdx.x;
    float4* q = p + c * (kChunk / 16);
    for (int i = threadIdx.x; i < kChunk / 16; i += 256) q[i] = make_float4(i, c, 1.f, 2.f);
}
__global__ __launch_bounds__(256) void rd(const float4* p, long nchunks, int rev, float* sink) {
    const long c = rev ? nchunks - 1 - blockIdx.x : blockIdx.x;
    const float4* q = p + c * (kChunk / 16);
    float4 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = q[threadIdx.x + k * 256];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += v[k].x + v[k].y + v[k].z + v[k].w;
    if (s == 12345.678f) sink[0] = s;
}
// read a, write b (a streaming pass): ascending or descending
__global__ __launch_bounds__(256) void cp(const float4* a, float4* b, long nchunks, int rev) {
    const long c = rev ? nchunks - 1 - blockIdx.x : blockIdx.x;
    const float4* q = a + c * (kChunk / 16);
    float4* o = b + c * (kChunk / 16);
    float4 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = q[threadIdx.x + k * 256];
#pragma unroll
    for (int k = 0; k < 16; ++k) { v[k].x = v[k].x * 2.f + 1.f; o[threadIdx.x + k * 256] = v[k]; }
}
int main() {
    float* sink; CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long sizes_mb[] = {64, 105, 160, 210, 320, 420, 840};
    printf("| MB | read after write, ascending GB/s | descending GB/s | chain of 6 passes (a->b->a...), all ascending ms | alternating ms |\n|---:|---:|---:|---:|---:|\n");
    for (long mb : sizes_mb) {
        const long n = mb << 20, nchunks = n / kChunk;
        float4 *a, *b; CK(hipMalloc(&a, n)); CK(hipMalloc(&b, n));
        float best[2] = {1e9f, 1e9f};
        for (int rep = 0; rep < 5; ++rep)
            for (int rev = 0; rev < 2; ++rev) {
                wr<<<dim3(nchunks), dim3(256)>>>(a, nchunks, 0);
                CK(hipEventRecord(e0));
                rd<<<dim3(nchunks), dim3(256)>>>(a, nchunks, rev, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best[rev]) best[rev] = ms;
            }
        float chain[2] = {1e9f, 1e9f};
        for (int rep = 0; rep < 5; ++rep)
            for (int alt = 0; alt < 2; ++alt) {
                wr<<<dim3(nchunks), dim3(256)>>>(a, nchunks, 0);
                CK(hipEventRecord(e0));
                for (int k = 0; k < 6; ++k) cp<<<dim3(nchunks), dim3(256)>>>(k & 1 ? b : a, k & 1 ? a : b, nchunks, alt ? !(k & 1) : 0);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < chain[alt]) chain[alt] = ms;
            }
        printf("| %ld | %.0f | %.0f | %.3f | %.3f |\n", mb, n / best[0] * 1e-6, n / best[1] * 1e-6, chain[0], chain[1]);
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
