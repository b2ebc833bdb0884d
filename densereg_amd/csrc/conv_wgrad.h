// conv_wgrad.h -- weight gradient of the stride-1 SAME convolutions on the fp32 matrix cores.
//
//   dW[tap][ci][co] = sum over pixels m of  x[m shifted by tap][ci] * g[m][co]
//
// GEMM view per tap: M = Cin, N = Cout, K = B*H*W pixels (the long axis).  Both operands are stored
// pixel-major with channels contiguous, which is exactly the k-major LDS image the 32x32x2 fp32 MFMA
// wants (A[i=l&31][k=l>>5] = Xs[pixel k][channel i]): tiles go HBM -> LDS as float4 rows, no transpose.
// The pixel axis is split over `nsplit` workgroups; each writes its partial HWIO tile to a scratch
// slab and a second kernel folds the slabs into the flat gradient accumulator -- deterministic, no
// floating-point atomics.  Replaces the Conv2DBackpropFilter ops TF derives for ops.py:282.
#pragma once
#include "dr_platform.h"

namespace dr {

struct WgradParams {
    const float* x; int x_cs; int x_coff; int Cin;
    const float* g; int g_cs; int g_coff; int Cout;
    int B, H, W, ksize;
    const float* rowmask; float mask_thresh;     // forward-input rows that were read as zero
    float* partial;                              // [nsplit][taps][Cin][Cout]
    int nsplit; int rows_per_split;              // multiple of 16
};

template <int T>   // tile T x T channels, 4 waves as 2 x 2
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
    constexpr int BK = 16;                 // pixels per step
    constexpr int ST = T + 4;              // LDS row stride (keeps float4 alignment)
    constexpr int WT = T / 2;              // wave tile
    constexpr int TM = WT / 32;
    constexpr int ITERS = (BK * (T / 4)) / 256;      // float4 loads per thread per operand
    static_assert(ITERS >= 1, "tile too small for the loader mapping");
    __shared__ float Xs[2][BK][ST];
    __shared__ float Gs[2][BK][ST];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lk = lane >> 5, li = lane & 31;
    const int co_tiles = dr_ceil_div(p.Cout, T);
    const int ci0 = (blockIdx.x / co_tiles) * T;
    const int co0 = (blockIdx.x % co_tiles) * T;
    const int tap = blockIdx.y;
    const int split = blockIdx.z;
    const int taps = p.ksize * p.ksize;
    const int pad = p.ksize / 2;
    const int dy = tap / p.ksize - pad, dx = tap % p.ksize - pad;
    const int HW = p.H * p.W;
    const long M = (long)p.B * HW;
    const long m_begin = (long)split * p.rows_per_split;
    const long m_end = m_begin + p.rows_per_split < M ? m_begin + p.rows_per_split : M;
    const int steps = m_begin < m_end ? (int)((m_end - m_begin + BK - 1) / BK) : 0;

    float4 xr[ITERS], gr[ITERS];
    int xnv[ITERS], gnv[ITERS];        // valid leading components of the staged float4 (0 = nothing)
    // unconditional loads + zero-select at store time (see conv_igemm.h: a branchy load makes hipcc wait right
    // behind every load and serialises the refill in front of the MFMAs)
    auto load = [&](int st) {
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / (T / 4);
            const int c4 = (idx % (T / 4)) * 4;
            const long m = m_begin + (long)st * BK + row;
            const bool in_range = m < m_end;
            // gradient row
            const int gleft = p.Cout - (co0 + c4);
            int gv = gleft < 0 ? 0 : (gleft > 4 ? 4 : gleft);
            if (!in_range) gv = 0;
            // shifted input row
            const int xleft = p.Cin - (ci0 + c4);
            int xv = xleft < 0 ? 0 : (xleft > 4 ? 4 : xleft);
            long ms = m;
            bool ok = in_range;
            if (p.ksize > 1) {
                const int rem = (int)((in_range ? m : 0) % HW);
                const int yy = rem / p.W + dy, xx = rem % p.W + dx;
                ok = ok && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                ms = m + (long)dy * p.W + dx;
            }
            if (!ok) { xv = 0; ms = 0; }
            if (p.rowmask && xv && p.rowmask[ms] < p.mask_thresh) xv = 0;
            const float* gsrc = gv ? p.g + m * p.g_cs + p.g_coff + co0 + c4 : p.g;
            const float* xsrc = xv ? p.x + ms * p.x_cs + p.x_coff + ci0 + c4 : p.x;
            gr[i] = *reinterpret_cast<const float4*>(gsrc);
            xr[i] = *reinterpret_cast<const float4*>(xsrc);
            gnv[i] = gv;
            xnv[i] = xv;
        }
    };
    auto zsel = [](float4 v, int nv) {
        return make_float4(nv > 0 ? v.x : 0.f, nv > 1 ? v.y : 0.f, nv > 2 ? v.z : 0.f, nv > 3 ? v.w : 0.f);
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / (T / 4);
            const int c4 = (idx % (T / 4)) * 4;
            *reinterpret_cast<float4*>(&Xs[buf][row][c4]) = zsel(xr[i], xnv[i]);
            *reinterpret_cast<float4*>(&Gs[buf][row][c4]) = zsel(gr[i], gnv[i]);
        }
    };

    dr_f32x16 acc[TM][TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (steps > 0) {
        load(0);
        store(0);
    }
    __syncthreads();
    for (int st = 0; st < steps; ++st) {
        const int buf = st & 1;
        if (st + 1 < steps) load(st + 1);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[TM], b[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = Xs[buf][2 * kk + lk][wm * WT + i * 32 + li];
#pragma unroll
            for (int j = 0; j < TM; ++j) b[j] = Gs[buf][2 * kk + lk][wn * WT + j * 32 + li];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (st + 1 < steps) store(buf ^ 1);
        __syncthreads();
    }

    // partial[split][tap][ci][co]; D: row(ci) = (r&3)+8*(r>>2)+4*lk, col(co) = li
    float* dst = p.partial + ((long)split * taps + tap) * p.Cin * p.Cout;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int co = co0 + wn * WT + j * 32 + li;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + wm * WT + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (ci < p.Cin && co < p.Cout) dst[(long)ci * p.Cout + co] = acc[i][j][r];
            }
    }
}

// dst[i] += sum_s partial[s][i].  block = 64 elements x 4 split lanes, folded through LDS.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* partial, int nsplit, long n, float* dst) {
    __shared__ float red[4][64];
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    for (long base = (long)blockIdx.x * 64; base < n; base += (long)gridDim.x * 64) {
        const long i = base + e;
        float s = 0.f;
        if (i < n)
            for (int k = sl; k < nsplit; k += 4) s += partial[(long)k * n + i];
        red[sl][e] = s;
        __syncthreads();
        if (sl == 0 && i < n) dst[i] += (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
        __syncthreads();
    }
}

// Stem (7x7/s2, Cin = 1): dW[ky][kx][n] += sum_pix x[b, oy*s+ky-pt, ox*s+kx-pl] * g[pix][n], Cout = 32.
// block = 256 threads = 8 tap groups x 32 channels over a chunk of 256 output pixels.
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* x, int B, int H, int W, const float* g, int g_cs, int k,
                                                         int stride, int pad_t, int pad_l, int Ho, int Wo, float* dw) {
    const int n = threadIdx.x & 31, tg = threadIdx.x >> 5;
    const long M = (long)B * Ho * Wo;
    const long m0 = (long)blockIdx.x * 256;
    float acc[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) acc[i] = 0.f;
    for (long m = m0; m < m0 + 256 && m < M; ++m) {
        const int b = (int)(m / ((long)Ho * Wo));
        const int rem = (int)(m % ((long)Ho * Wo));
        const int oy = rem / Wo, ox = rem % Wo;
        const float gv = g[m * g_cs + n];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int tap = tg + 8 * i;
            if (tap < k * k) {
                const int iy = oy * stride + tap / k - pad_t, ix = ox * stride + tap % k - pad_l;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) acc[i] = fmaf(x[((long)b * H + iy) * W + ix], gv, acc[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int tap = tg + 8 * i;
        if (tap < k * k) atomicAdd(&dw[tap * 32 + n], acc[i]);
    }
}

}  // namespace dr
