// conv_wgrad.h -- weight gradient of the stride-1 SAME convolutions on the fp32 matrix cores.
//
//   dW[tap][ci][co] = sum over pixels m of  x[m shifted by tap][ci] * g[m][co]
//
// GEMM view per tap: M = Cin, N = Cout, K = B*H*W pixels (the long axis).  Both operands are stored
// pixel-major with channels contiguous, which is exactly the k-major LDS image the 32x32x2 fp32 MFMA
// wants (A[i=l&31][k=l>>5] = Xs[pixel k][channel i]): tiles go HBM -> LDS as float4 rows, no transpose.
// Tensors are addressed with 32-bit element offsets (the host rejects anything larger).
// The pixel axis is split over `nsplit` workgroups; each writes its partial HWIO tile to a scratch
// slab and a second kernel folds the slabs into the flat gradient accumulator -- deterministic, no
// floating-point atomics.  Replaces the Conv2DBackpropFilter ops TF derives for ops.py:282.
#pragma once
#include <type_traits>

#include "dr_platform.h"

namespace dr {

struct WgradParams {
    const float* x; int x_cs; int x_coff; int Cin;
    const float* g; int g_cs; int g_coff; int Cout;
    int B, H, W, ksize;
    const float* rowmask; float mask_thresh;     // forward-input rows that were read as zero
    float* partial;                              // [nsplit][taps][Cin][Cout]
    int nsplit; int rows_per_split;              // multiple of 16
    int x_bf16;                                  // conv_wgrad_tr_kernel only: x holds bf16 elements (BnTrainParams::out_bf16)
    int g_bf16;                                  // conv_wgrad_bf16_kernel only: g holds bf16 elements (stride g_cs elements, channel
                                                 // groups of four zero-padded) -- see BnBwdParams::draw_bf16
    // A layer whose Cin is a multiple of 128 plus a few channels (the comb|uvd inputs of the heads: 515 = 512 + u, v, d; 131 = 128 + 3)
    // runs as TWO launches into the same slabs: conv_wgrad_x3_kernel<128> on the leading cin_total - tail channels (Cin = that), and
    // conv_wgrad_tail_kernel on the last ones (ci_base = their first channel).  cin_total = rows of a slab [Cin rows][Cout]; 0 = Cin.
    int cin_total, ci_base;
};

// Weight gradient of a FEW input channels (<= 4) of a 1x1 layer: dW[ci][co] = sum_pix x[pix][ci_base + ci] * g[pix][co].  A workgroup
// = 64 output channels x its slab of pixels; lane = output channel, the four waves take the pixels m = wave (mod 4) (fp32 fma chains
// in pixel order, the arithmetic class of the fp32-MFMA kernels) and are summed in wave order through LDS: deterministic.  g is
// streamed once, coalesced (256 B per wave and pixel) -- 0.6 GFLOP and 420 MB for 515 -> 512 at 204 800 pixels: HBM-bound, where the
// square-tile kernel spent a 128-channel tile (or nine 64-channel ones) on three channels.  Grid = nsplit x ceil(Cout / 64).
__global__ __launch_bounds__(256) void conv_wgrad_tail_kernel(const WgradParams p) {
    __shared__ float red[4][4][64];
    const int split = blockIdx.x % p.nsplit, cblk = blockIdx.x / p.nsplit;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int co = cblk * 64 + lane;
    const int M = p.B * p.H * p.W;
    const int m_begin = split * p.rows_per_split;
    const int m_end = m_begin + p.rows_per_split < M ? m_begin + p.rows_per_split : M;
    const int nt = p.Cin;                                      // tail channels (1..4)
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const bool live = co < p.Cout;
    const float* gp = p.g + p.g_coff + (live ? co : 0);
    const float* xp = p.x + p.x_coff + p.ci_base;
    constexpr int U = 8;                                       // pixels in flight per wave
    for (int m0 = m_begin + wave; m0 < m_end; m0 += 4 * U) {
        float gv[U], xv[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int mu = m0 + 4 * u;
            const int m = mu < m_end ? mu : m_begin;
            gv[u] = gp[(long)m * p.g_cs];
            bool on = mu < m_end;
            if (on && p.rowmask) on = !(p.rowmask[m] < p.mask_thresh);
#pragma unroll
            for (int k = 0; k < 4; ++k) xv[u][k] = (on && k < nt) ? xp[(long)m * p.x_cs + k] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = fmaf(xv[u][k], gv[u], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[wave][k][lane] = acc[k];
    __syncthreads();
    if (wave == 0 && live) {
        const int ct = p.cin_total ? p.cin_total : p.Cin;
        float* dst = p.partial + (long)split * ct * p.Cout;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < nt) dst[(long)(p.ci_base + k) * p.Cout + co] = ((red[0][k][lane] + red[1][k][lane]) + red[2][k][lane]) + red[3][k][lane];
    }
}

// Tile T x T channels, 4 waves as 2 x 2, wave tile T/2 x T/2.
// T = 128: a wave owns 64 x 64 = 2 x 2 MFMA tiles of 32 contiguous channels each (two dwords per operand and k-step,
// fused by hipcc into one ds_read2_b32); tiles entirely beyond Cin / Cout are skipped.  (A channel-pair mapping
// read with one ds_read_b64 was measured equal on full tiles and cannot skip anything on ragged ones.)
// T = 64: one MFMA tile per wave.
// The kernel body, shared by the one-layer launch (conv_wgrad_kernel) and the grouped launch of many small layers
// (conv_wgrad_group_kernel): `bid` is the workgroup's index within ITS layer's grid.
template <int T>
__device__ __forceinline__ void conv_wgrad_body(const WgradParams& p, const int bid) {
    constexpr int BK = 16;                 // pixels per step
    constexpr int ST = T + 4;              // LDS row stride (keeps float4 alignment)
    constexpr int WT = T / 2;              // wave tile
    constexpr int TM = WT / 32;
    constexpr int ITERS = (BK * (T / 4)) / 256;      // float4 loads per thread per operand
    constexpr int RSTEP = 256 / (T / 4);             // pixel rows between a thread's successive loads
    static_assert(ITERS >= 1 && 256 % (T / 4) == 0, "tile too small for the loader mapping");
    __shared__ __attribute__((aligned(16))) float Xs[2][BK][ST];
    __shared__ __attribute__((aligned(16))) float Gs[2][BK][ST];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lk = lane >> 5, li = lane & 31;
    // 1-D grid of tiles * taps * nsplit workgroups, dealt round-robin to the 8 XCDs (id % 8).  Every (tile, tap) of
    // one pixel slab reads the same x / g rows: with nsplit % 8 == 0 slab s lives on XCD s % 8, so each slab is
    // fetched from HBM by one L2 instead of by all eight.
    const int co_tiles = dr_ceil_div(p.Cout, T);
    const int taps = p.ksize * p.ksize;
    const int tiles = dr_ceil_div(p.Cin, T) * co_tiles;
    int split, rest;
    if ((p.nsplit & 7) == 0) {
        const int per = p.nsplit >> 3, j = bid >> 3;
        split = (j % per) * 8 + (bid & 7);
        rest = j / per;
    } else {
        split = bid % p.nsplit;
        rest = bid / p.nsplit;
    }
    const int tile = rest % tiles, tap = rest / tiles;
    const int ci0 = (tile / co_tiles) * T;
    const int co0 = (tile % co_tiles) * T;
    const int pad = p.ksize / 2;
    const int dy = tap / p.ksize - pad, dx = tap % p.ksize - pad;
    const int HW = p.H * p.W;
    const int M = p.B * HW;
    const int m_begin = split * p.rows_per_split;
    const int m_end = m_begin + p.rows_per_split < M ? m_begin + p.rows_per_split : M;
    const int steps = m_begin < m_end ? (m_end - m_begin + BK - 1) / BK : 0;
    // image sides are powers of two on this network (2..64): shifts instead of divisions in the loader
    const bool pow2 = (p.W & (p.W - 1)) == 0 && (HW & (HW - 1)) == 0;
    const int w_shift = __builtin_ctz((unsigned)p.W);

    // ---- loader: everything that does not change from step to step is computed once -------------
    // A thread always loads the same 4 channels (c4) of pixel rows row0 + i*RSTEP of every step.  Loads are
    // unconditional (a predicated-off lane reads the tensor base, its value is dropped at store time) and use
    // 32-bit element offsets; the optional row mask travels with the prefetch instead of gating it.
    const int c4 = (tid % (T / 4)) * 4;
    const int row0 = tid / (T / 4);
    const int gleft = p.Cout - (co0 + c4), xleft = p.Cin - (ci0 + c4);
    const int g_nvc = gleft < 0 ? 0 : (gleft > 4 ? 4 : gleft);       // valid components of this thread's float4
    const int x_nvc = xleft < 0 ? 0 : (xleft > 4 ? 4 : xleft);
    const unsigned g_base = (unsigned)(p.g_coff + co0 + c4), x_base = (unsigned)(p.x_coff + ci0 + c4);
    const int tap_shift = dy * p.W + dx;

    float4 xr[ITERS], gr[ITERS];
    float xm[ITERS];
    int xnv[ITERS], gnv[ITERS];        // valid leading components of the staged float4 (0 = nothing)
    int next_step = 0;
    auto load = [&]() __attribute__((always_inline)) {
        const int mb = m_begin + next_step * BK + row0;
        ++next_step;
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int m = mb + i * RSTEP;
            const bool in_range = m < m_end;
            bool ok = in_range;
            if (p.ksize > 1) {
                const int mm = in_range ? m : 0;
                int yy, xx;
                if (pow2) {
                    const int rem = mm & (HW - 1);
                    yy = (rem >> w_shift) + dy; xx = (rem & (p.W - 1)) + dx;
                } else {
                    const int rem = mm % HW;
                    yy = rem / p.W + dy; xx = rem % p.W + dx;
                }
                ok = ok && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            }
            const unsigned ms = ok ? (unsigned)(m + tap_shift) : 0u;
            const int gv = in_range ? g_nvc : 0, xv = ok ? x_nvc : 0;
            gr[i] = *reinterpret_cast<const float4*>(gv ? p.g + ((unsigned)m * (unsigned)p.g_cs + g_base) : p.g);
            xr[i] = *reinterpret_cast<const float4*>(xv ? p.x + (ms * (unsigned)p.x_cs + x_base) : p.x);
            if (p.rowmask) xm[i] = p.rowmask[ms];
            gnv[i] = gv;
            xnv[i] = xv;
        }
    };
    auto zsel = [](float4 v, int nv) {
        return make_float4(nv > 0 ? v.x : 0.f, nv > 1 ? v.y : 0.f, nv > 2 ? v.z : 0.f, nv > 3 ? v.w : 0.f);
    };
    auto store = [&](const int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int row = row0 + i * RSTEP;
            const int xv = (p.rowmask && xm[i] < p.mask_thresh) ? 0 : xnv[i];
            *reinterpret_cast<float4*>(&Xs[buf][row][c4]) = zsel(xr[i], xv);
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
        load();
        store(0);
    }
    __syncthreads();
    // live 32x32 MFMA tiles of this wave along ci / co (0..TM)
    const int na_ = (p.Cin - (ci0 + wm * WT) + 31) / 32, nb_ = (p.Cout - (co0 + wn * WT) + 31) / 32;
    const int na = na_ < 0 ? 0 : (na_ > TM ? TM : na_), nb = nb_ < 0 ? 0 : (nb_ > TM ? TM : nb_);
    // one step of 16 pixels from LDS buffer `buf` (a compile-time constant: the loop below is unrolled by two)
    auto k_step = [&](const int buf, const bool more) __attribute__((always_inline)) {
        if (more) load();
        if constexpr (TM == 2) {
            float a[BK / 2][2], b[BK / 2][2];                    // every fragment of the step, read up front
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    a[kk][t] = Xs[buf][2 * kk + lk][wm * WT + 32 * t + li];
                    b[kk][t] = Gs[buf][2 * kk + lk][wn * WT + 32 * t + li];
                }
            }
            // MFMA tiles that lie entirely beyond Cin / Cout are skipped (wave-uniform): a 78x78 layer on this 128x128
            // tile has 3x3 live 32x32 tiles out of 4x4 -- the kernel is MFMA-bound, so that is 44 % of its time
            auto mf = [&](auto NA, auto NB) __attribute__((always_inline)) {
#pragma unroll
                for (int kk = 0; kk < BK / 2; ++kk)
#pragma unroll
                    for (int i = 0; i < decltype(NA)::value; ++i)
#pragma unroll
                        for (int j = 0; j < decltype(NB)::value; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
            };
            using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>;
            if (na == 2) {
                if (nb == 2) mf(I2{}, I2{});
                else if (nb == 1) mf(I2{}, I1{});
            } else if (na == 1) {
                if (nb == 2) mf(I1{}, I2{});
                else if (nb == 1) mf(I1{}, I1{});
            }
        } else {
            float a[BK / 2], b[BK / 2];
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                a[kk] = Xs[buf][2 * kk + lk][wm * WT + li];
                b[kk] = Gs[buf][2 * kk + lk][wn * WT + li];
            }
            if (na > 0 && nb > 0) {
#pragma unroll
                for (int kk = 0; kk < BK / 2; ++kk)
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[kk], acc[0][0], 0, 0, 0);
            }
        }
        if (more) store(buf ^ 1);
        __syncthreads();
    };
    const int pairs = steps & ~1;
    for (int st = 0; st < pairs; st += 2) {
        k_step(0, true);
        k_step(1, st + 2 < steps);
    }
    if (steps & 1) k_step(0, false);

    // partial[split][tap][ci][co]; D: row = (r&3)+8*(r>>2)+4*lk, col = li; MFMA tile t = channels [32t, 32t+32)
    float* dst = p.partial + ((long)split * taps + tap) * p.Cin * p.Cout;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int co = co0 + wn * WT + 32 * j + li;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
                const int ci = ci0 + wm * WT + 32 * i + row;
                if (ci < p.Cin && co < p.Cout) dst[(long)ci * p.Cout + co] = acc[i][j][r];
            }
    }
}

template <int T>
__global__ __launch_bounds__(256, (T == 128 ? 3 : 6)) void conv_wgrad_kernel(const WgradParams p) {
    conv_wgrad_body<T>(p, (int)blockIdx.x);
}

// Grouped launch: the weight gradients of MANY small layers in one grid.  Everything below 32x32 pixels is a chain of
// launches that cannot fill the chip (a 3x3 64->64 layer at 8x8 is 2560 pixels: its own launch takes ~9 us whatever the
// split); their weight gradients are not on the critical path of the backward sweep -- only the fold at its end needs
// them -- so the executor keeps each such layer's G tensor alive, collects the layers in a table and runs all of them
// as one launch right before the slab fold.  Segment s owns workgroups [first_block[s], first_block[s+1]).
struct WgradGroupSeg { WgradParams p; int first_block; int pad_; };
__global__ __launch_bounds__(256, 6) void conv_wgrad_group_kernel(const WgradGroupSeg* segs, int nseg) {
    int lo = 0, hi = nseg - 1;                                  // last segment whose first_block <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (segs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const WgradParams p = segs[lo].p;
    conv_wgrad_body<64>(p, (int)blockIdx.x - segs[lo].first_block);
}

// ------------------------------------------------------------------------------------------------------------
// 3x3 layers with 65..96 channels on BOTH sides (the hm3 / um-head residuals: 65->65, 78->78): on the square tiles a
// layer like that is nine single-tile workgroups per slab, each re-reading x and g and each with a 128x128 tile that
// is 9/16 live -- 37 TFLOP/s, latency-bound.  Here ONE workgroup owns a kernel ROW (dy) of a slab: the three taps
// dx = -1, 0, +1 share the G tile, X is staged once per dx (each with its own border mask, like a tap of the square
// kernel), and the 27 = 3 taps x 3 x 3 MFMA tiles are dealt round-robin to the four waves (7/7/7/6).
// ------------------------------------------------------------------------------------------------------------
// (no run-time tile predicates in here: a branch per MFMA serialises every LDS read behind its own wait -- measured
// 4 us per step instead of 1.5; the host only selects this kernel when all 3 x 3 tiles are live)
template <int W>
__device__ __forceinline__ void wgrad_row_mfma(const float (*Xs)[16][100], const float (*Gs)[100], int lk, int li,
                                               dr_f32x16 (&acc)[7]) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        float a[3][3], b[3];                                  // every fragment of the k-step (unused ones are dropped)
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int t = 0; t < 3; ++t) a[d][t] = Xs[d][2 * kk + lk][32 * t + li];
#pragma unroll
        for (int t = 0; t < 3; ++t) b[t] = Gs[2 * kk + lk][32 * t + li];
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const int t = W + 4 * q;                          // compile-time after unrolling
            if (t < 27) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t / 9][(t % 9) / 3], b[t % 3], acc[q], 0, 0, 0);
        }
    }
}

__global__ __launch_bounds__(256, 2) void conv_wgrad_row_kernel(const WgradParams p) {
    constexpr int BK = 16, CT = 96, ST = CT + 4, C4 = CT / 4;     // 24 float4 per pixel row
    constexpr int XI = (3 * BK * C4 + 255) / 256;                 // 5 float4 of X per thread and step (the last partly)
    constexpr int GI = (BK * C4 + 255) / 256;                     // 2 of G
    __shared__ __attribute__((aligned(16))) float Xs[2][3][BK][ST];
    __shared__ __attribute__((aligned(16))) float Gs[2][BK][ST];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lk = lane >> 5, li = lane & 31;
    int split, dyi;
    if ((p.nsplit & 7) == 0) {                                    // slab s on XCD s % 8 (see conv_wgrad_kernel)
        const int per = p.nsplit >> 3, j = blockIdx.x >> 3;
        split = (j % per) * 8 + (blockIdx.x & 7);
        dyi = j / per;
    } else {
        split = blockIdx.x % p.nsplit;
        dyi = blockIdx.x / p.nsplit;
    }
    const int dy = dyi - 1;
    const int HW = p.H * p.W;
    const int M = p.B * HW;
    const int m_begin = split * p.rows_per_split;
    const int m_end = m_begin + p.rows_per_split < M ? m_begin + p.rows_per_split : M;
    const int steps = m_begin < m_end ? (m_end - m_begin + BK - 1) / BK : 0;
    const bool pow2 = (p.W & (p.W - 1)) == 0 && (HW & (HW - 1)) == 0;
    const int w_shift = __builtin_ctz((unsigned)p.W);

    float4 xr[XI], gr[GI];
    float xm[XI];
    int xnv[XI], gnv[GI];
    int next_step = 0;
    auto load = [&]() __attribute__((always_inline)) {
        const int mb = m_begin + next_step * BK;
        ++next_step;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int idx = tid + i * 256;
            const int dxi = idx / (BK * C4), rem = idx % (BK * C4);
            const int row = rem / C4, c4 = (rem % C4) * 4;
            const int m = mb + row;
            const int left = p.Cin - c4;
            int nv = left < 0 ? 0 : (left > 4 ? 4 : left);
            bool ok = idx < 3 * BK * C4 && m < m_end && nv > 0;
            const int mm = ok ? m : 0;
            int y, x;
            if (pow2) { const int r = mm & (HW - 1); y = r >> w_shift; x = r & (p.W - 1); }
            else { const int r = mm % HW; y = r / p.W; x = r % p.W; }
            const int yy = y + dy, xx = x + dxi - 1;
            ok = ok && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            const unsigned ms = ok ? (unsigned)(m + dy * p.W + dxi - 1) : 0u;
            xr[i] = *reinterpret_cast<const float4*>(ok ? p.x + (ms * (unsigned)p.x_cs + (unsigned)(p.x_coff + c4)) : p.x);
            if (p.rowmask) xm[i] = p.rowmask[ms];
            xnv[i] = ok ? nv : 0;
        }
#pragma unroll
        for (int i = 0; i < GI; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / C4, c4 = (idx % C4) * 4;
            const int m = mb + row;
            const int left = p.Cout - c4;
            const int nv = left < 0 ? 0 : (left > 4 ? 4 : left);
            const bool ok = idx < BK * C4 && m < m_end && nv > 0;
            gr[i] = *reinterpret_cast<const float4*>(ok ? p.g + ((unsigned)m * (unsigned)p.g_cs + (unsigned)(p.g_coff + c4)) : p.g);
            gnv[i] = ok ? nv : 0;
        }
    };
    auto zsel = [](float4 v, int nv) {
        return make_float4(nv > 0 ? v.x : 0.f, nv > 1 ? v.y : 0.f, nv > 2 ? v.z : 0.f, nv > 3 ? v.w : 0.f);
    };
    auto store = [&](const int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int idx = tid + i * 256;
            if (idx < 3 * BK * C4) {
                const int dxi = idx / (BK * C4), rem = idx % (BK * C4);
                const int nv = (p.rowmask && xm[i] < p.mask_thresh) ? 0 : xnv[i];
                *reinterpret_cast<float4*>(&Xs[buf][dxi][rem / C4][(rem % C4) * 4]) = zsel(xr[i], nv);
            }
        }
#pragma unroll
        for (int i = 0; i < GI; ++i) {
            const int idx = tid + i * 256;
            if (idx < BK * C4) *reinterpret_cast<float4*>(&Gs[buf][idx / C4][(idx % C4) * 4]) = zsel(gr[i], gnv[i]);
        }
    };

    dr_f32x16 acc[7];
#pragma unroll
    for (int q = 0; q < 7; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    if (steps > 0) {
        load();
        store(0);
    }
    __syncthreads();
    auto k_step = [&](const int buf, const bool more) __attribute__((always_inline)) {
        if (more) load();
        switch (wave) {                                           // per-wave tile list is compile-time inside each case
            case 0: wgrad_row_mfma<0>(Xs[buf], Gs[buf], lk, li, acc); break;
            case 1: wgrad_row_mfma<1>(Xs[buf], Gs[buf], lk, li, acc); break;
            case 2: wgrad_row_mfma<2>(Xs[buf], Gs[buf], lk, li, acc); break;
            default: wgrad_row_mfma<3>(Xs[buf], Gs[buf], lk, li, acc); break;
        }
        if (more) store(buf ^ 1);
        __syncthreads();
    };
    const int pairs = steps & ~1;
    for (int st = 0; st < pairs; st += 2) {
        k_step(0, true);
        k_step(1, st + 2 < steps);
    }
    if (steps & 1) k_step(0, false);

    // partial[split][tap][ci][co], tap = dyi*3 + dxi
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        const int t = wave + 4 * q;
        if (t < 27) {
            const int dxi = t / 9, ti = (t % 9) / 3, tj = t % 3;
            float* dst = p.partial + ((long)split * 9 + dyi * 3 + dxi) * p.Cin * p.Cout;
            const int co = 32 * tj + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (ci < p.Cin && co < p.Cout) dst[(long)ci * p.Cout + co] = acc[q][r];
            }
        }
    }
}

// dst[i] += sum_s partial[s][i].  block = 64 elements x 4 split lanes, folded through LDS in a fixed order.
// The slab loads of a lane are issued in batches of 8 independent loads (a plain accumulate loop waits for each).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* partial, int nsplit, long n, float* dst) {
    __shared__ float red[4][64];
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    for (long base = (long)blockIdx.x * 64; base < n; base += (long)gridDim.x * 64) {
        const long i = base + e;
        float s = 0.f;
        if (i < n) {
            int k = sl;
            for (; k + 7 * 4 < nsplit; k += 8 * 4) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = partial[(long)(k + u * 4) * n + i];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; k < nsplit; k += 4) s += partial[(long)k * n + i];
        }
        red[sl][e] = s;
        __syncthreads();
        if (sl == 0 && i < n) dst[i] += (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
        __syncthreads();
    }
}

// Deferred fold of EVERY layer's slabs in one launch at the end of the backward sweep (145 separate ~5 us fold
// launches otherwise).  Segment s covers workgroups [first_block[s], first_block[s+1]) of 64 elements each.
struct FoldSeg { long dst_off; long n; long slab_off; int nsplit; int first_block; };
__global__ __launch_bounds__(256) void wgrad_fold_all_kernel(const float* slabs, const FoldSeg* segs, int nseg, float* grad) {
    __shared__ float red[4][64];
    int lo = 0, hi = nseg - 1;                                  // last segment whose first_block <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (segs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const FoldSeg sg = segs[lo];
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long i = (long)((int)blockIdx.x - sg.first_block) * 64 + e;
    const float* part = slabs + sg.slab_off;
    float s = 0.f;
    if (i < sg.n) {
        int k = sl;
        for (; k + 7 * 4 < sg.nsplit; k += 8 * 4) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = DR_NT_LOAD(8, &part[(long)(k + u * 4) * sg.n + i]);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < sg.nsplit; k += 4) s += DR_NT_LOAD(8, &part[(long)k * sg.n + i]);
    }
    red[sl][e] = s;
    __syncthreads();
    if (sl == 0 && i < sg.n) grad[sg.dst_off + i] += (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

// Stem (um_v1.py:86: 7x7 / stride 2, Cin = 1, Cout = 32): dW[ky][kx][n] = sum_pix x[b, 2*oy+ky-pt, 2*ox+kx-pl] * g[pix][n].
// A unit is 64 consecutive output pixels of one output row.  The workgroup stages the unit's input window (7 rows x 133
// columns, zero outside the image) and its 64 x 32 gradient rows in LDS once; thread (ky, n) then walks the 64 pixels
// with a sliding 7-wide register window over row ky -- per pixel two new LDS words (broadcast within the 32 lanes of a
// ky) and one gradient word feed 7 FMAs.  (The previous version issued 8 global loads per 7 FMAs: 221 us for 0.5 GFLOP.)
// A workgroup accumulates `units_per_wg` units in registers and writes one partial row [7*7][32] (tap-major) that
// wgrad_reduce_kernel folds into the flat gradient -- no floating-point atomics.
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* x, int B, int H, int W, const float* g, int g_cs, int pad_t,
                                                         int pad_l, int Ho, int Wo, int units_per_wg, float* partial) {
    constexpr int K = 7, S = 2, PX = 64, XW = S * (PX - 1) + K;          // 133 input columns under 64 output pixels
    __shared__ float xs[K][XW + 3];
    __shared__ __attribute__((aligned(16))) float gs[PX][32];
    const int tid = threadIdx.x, n = tid & 31, ky = tid >> 5;             // ky = 7: staging only
    const int nseg = Wo / PX;
    const int units = B * Ho * nseg;
    float acc[K];
#pragma unroll
    for (int kx = 0; kx < K; ++kx) acc[kx] = 0.f;
    for (int uu = 0; uu < units_per_wg; ++uu) {
        const int u = blockIdx.x * units_per_wg + uu;
        if (u >= units) break;                                            // uniform
        const int seg = u % nseg, r = u / nseg, oy = r % Ho, b = r / Ho;
        const int ix0 = seg * PX * S - pad_l, iy0 = oy * S - pad_t;
        __syncthreads();                                                  // the previous unit's readers are done
        for (int i = tid; i < K * XW; i += 256) {
            const int yy = i / XW, xx = i - yy * XW;
            const int iy = iy0 + yy, ix = ix0 + xx;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const float v = x[ok ? ((long)b * H + iy) * W + ix : 0];
            xs[yy][xx] = ok ? v : 0.f;
        }
        const long m0 = (long)r * Wo + seg * PX;
        for (int i = tid; i < PX * 8; i += 256) {
            const int px = i >> 3, c4 = (i & 7) * 4;
            *reinterpret_cast<float4*>(&gs[px][c4]) = *reinterpret_cast<const float4*>(g + (m0 + px) * g_cs + c4);
        }
        __syncthreads();
        if (ky < K) {
            float win[K + 2];
#pragma unroll
            for (int kx = 0; kx < K - 2; ++kx) win[kx] = xs[ky][kx];
#pragma unroll
            for (int ox = 0; ox < PX; ++ox) {
                win[K - 2] = xs[ky][S * ox + K - 2];
                win[K - 1] = xs[ky][S * ox + K - 1];
                const float gv = gs[ox][n];
#pragma unroll
                for (int kx = 0; kx < K; ++kx) acc[kx] = fmaf(win[kx], gv, acc[kx]);
#pragma unroll
                for (int kx = 0; kx < K - 2; ++kx) win[kx] = win[kx + 2];
            }
        }
    }
    if (ky < K) {
        float* row = partial + (long)blockIdx.x * K * K * 32;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) row[(ky * K + kx) * 32 + n] = acc[kx];
    }
}

}  // namespace dr
