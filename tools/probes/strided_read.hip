// How fast does HBM deliver a [M][C] fp32 tensor when it is read the way the 1x1 conv kernels read their pixels -- 128-row workgroups,
// one 64-byte piece of every row per K-tile, the next piece a K-tile later -- against reading the same rows contiguously?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/strided_read.hip -o tools/probes/bin/strided_read && tools/probes/bin/strided_read
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
// conv-like: workgroup = 128 rows, 512 threads: thread (row = tid / 4, q = tid % 4) reads 16 bytes of K-tile t; INFLIGHT tiles are
// requested before the first is consumed; nblk column-block workgroups read the same rows (as the conv's N blocks do)
template <int INFLIGHT, int PIECE>   // PIECE: consecutive K-tiles fetched as one contiguous piece per row (1: 64 B, 2: 128 B, 4: 256 B)
__global__ __launch_bounds__(512, 4) void conv_like(const float4* x, int M, int C4, int gy, float* sink) {
    const int L = blockIdx.x;                       // linear id; row block = L / gy (the N blocks of a row block are consecutive)
    const int mblk = L / gy;
    const int tid = threadIdx.x;
    constexpr int TPR = 4 * PIECE;                  // threads per row
    constexpr int ROWS = 512 / TPR;                 // rows per pass
    const int KT = C4 / 4;                          // K-tiles (16 channels = 4 float4)
    float s = 0.f;
    for (int rp = 0; rp < 128; rp += ROWS) {
        const int row = mblk * 128 + rp + tid / TPR, q = tid % TPR;
        const float4* p = x + (long)row * C4 + q;
        for (int t = 0; t < KT; t += PIECE * INFLIGHT) {
            float4 v[INFLIGHT];
#pragma unroll
            for (int u = 0; u < INFLIGHT; ++u) v[u] = p[(t + u * PIECE) * 4 < C4 ? (t + u * PIECE) * 4 : 0];
#pragma unroll
            for (int u = 0; u < INFLIGHT; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
            if (PIECE == 1 && INFLIGHT == 1) __syncthreads();      // the conv's barrier per K-tile
        }
    }
    if (s == 12345.678f) sink[0] = s;
}
__global__ __launch_bounds__(512, 4) void rows_contig(const float4* x, int M, int C4, int gy, float* sink) {
    const int mblk = blockIdx.x / gy;
    const float4* p = x + (long)mblk * 128 * C4;
    float s = 0.f;
    for (int i = threadIdx.x; i < 128 * C4; i += 512 * 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = p[i + u * 512 < 128 * C4 ? i + u * 512 : 0];
#pragma unroll
        for (int u = 0; u < 4; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (s == 12345.678f) sink[0] = s;
}
__global__ __launch_bounds__(256) void fill(float4* x, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] = make_float4(i & 7, 1.f, 2.f, 3.f);
}
int main() {
    const int M = 204800;
    float* sink; CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("| C | N blocks | pattern | us | GB/s of the tensor's bytes (read once from HBM if the N blocks share) |\n|---:|---:|---|---:|---:|\n");
    for (int C : {512, 256, 128}) {
        const int C4 = C / 4;
        const long n4 = (long)M * C4;
        float4 *x, *y; CK(hipMalloc(&x, n4 * 16)); CK(hipMalloc(&y, 512l << 20));
        for (int gy : {1, 4}) {
            auto run = [&](const char* name, auto launch) -> int {
                float best = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    fill<<<dim3(4096), dim3(256)>>>(y, (512l << 20) / 16);       // push the tensor out of the caches
                    fill<<<dim3(4096), dim3(256)>>>(x, n4);                       // ... and write it, as the producing pass does
                    CK(hipEventRecord(e0));
                    launch();
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                printf("| %d | %d | %s | %.1f | %.0f |\n", C, gy, name, best * 1e3, n4 * 16.0 / best * 1e-6);
                return 0;
            };
            const dim3 grid(M / 128 * gy);
            run("rows contiguous", [&] { rows_contig<<<grid, dim3(512)>>>(x, M, C4, gy, sink); });
            run("64 B per row and K-tile, 1 in flight + barrier", [&] { conv_like<1, 1><<<grid, dim3(512)>>>(x, M, C4, gy, sink); });
            run("64 B per row and K-tile, 2 in flight", [&] { conv_like<2, 1><<<grid, dim3(512)>>>(x, M, C4, gy, sink); });
            run("64 B per row and K-tile, 4 in flight", [&] { conv_like<4, 1><<<grid, dim3(512)>>>(x, M, C4, gy, sink); });
            run("128 B pieces, 2 in flight", [&] { conv_like<2, 2><<<grid, dim3(512)>>>(x, M, C4, gy, sink); });
            run("256 B pieces, 2 in flight", [&] { conv_like<2, 4><<<grid, dim3(512)>>>(x, M, C4, gy, sink); });
        }
        CK(hipFree(x)); CK(hipFree(y));
    }
    return 0;
}
