// hg_fused.h -- the bottom of an hourglass (everything at 8x8 pixels and below) as ONE launch, eval mode.
//
// um_v1.py:51-69 recurses  pool -> residual -> [hourglass] -> residual -> upsample + add  down to 2x2 pixels.  In eval mode
// BatchReNorm is a folded scale | shift per channel (ops.py:173-180), nothing couples the crops of a batch, and below 16x16 a
// layer is a few thousand rows: every one of those launches is a 5-10 us dependent chain that cannot fill the chip, and the part of
// the hourglass below 16x16 is 29 of them (24 convolutions, 3 pools, 2 upsample-adds) per stack.  Here ONE workgroup takes ONE crop
// through all of it with every intermediate tensor in LDS (160 KB per CU on gfx950; 80 KB used at F = 128):
//
//     A  = pool(x @16x16)                       8x8 x F       x: the hourglass level's input in HBM
//     A  = res0(A)                               lower1 @8
//     C  = pool(A);  A = res1(A)                 4x4 x F;      upper1 @8   (a residual module is computed IN PLACE: out = f(in) + in)
//     C  = res2(C);  D = pool(C);  C = res3(C)   lower1 @4; 2x2 x F; upper1 @4
//     D  = res4(D);  D = res5(D)                 lower1 / lower3 @2
//     C += up(D);  C = res6(C)                   lower3 @4
//     A += up(C);  y = res7(A)                   lower3 @8 -> HBM (the level's upsample-add stays a launch of its own)
//
// A residual module (um_v1.py:18-48 with num_out = C: identity skip) is 1x1 C -> C/2, 3x3 C/2 -> C/2, 1x1 C/2 -> C, each followed
// by the folded BatchReNorm and ReLU, plus the skip.  Convolutions run on v_mfma_f32_16x16x4_f32 (exact fp32, like every other conv
// of the path): rows = pixels (one 16-row tile at 4x4 and 2x2, four at 8x8), columns = output channels, 16 per tile; wave w owns
// the column tiles w, w + 4 and ALL row tiles, so a weight fragment is fetched once per workgroup -- straight from the packed
// weights in HBM / L2 into registers (the forward packing [Kp/16][tap][Np][16] is exactly "four consecutive k of output channel
// n": one 16-byte load per lane and K-group, no LDS staging; all workgroups read the same weights at about the same time).
// Activation fragments come from the LDS image [pixel][channel] (row stride C + 4 floats: an odd number of 16-byte slots, so
// the 16 rows of a ds_read_b128 lane group fall on different banks); a 3x3 tap outside the image reads a row of zeros.
// With one row tile (4x4, 2x2) the four MFMA steps of a K-group feed four independent accumulators (summed in a fixed order).
#pragma once
#include "dr_platform.h"
#include "kernels_misc.h"

namespace dr {

struct HgConvDesc { const float* w; const float* scale; const float* shift; int Np; int pad_; };   // packed fp32 weights, folded BN
struct HgFusedParams {
    const float* x; int x_cs; int x_coff;          // [B][16][16] pixels, F channels at x_coff
    float* y; int y_cs; int y_coff;                // [B][8][8] pixels, F channels at y_coff
    int B, F;
    HgConvDesc conv[24];                           // residual modules 0..7 in the order above, three convolutions each
};

// floats of LDS the kernel needs for F channels
__host__ __device__ inline int hg_fused_lds_floats(int F) {
    const int sF = F + 4, sH = F / 2 + 4;
    return 64 * sF + 2 * 64 * sH + 16 * sF + 4 * sF + sF;
}
inline bool hg_fused_supported(int F) { return F >= 32 && F % 32 == 0 && F <= 128; }

// 3x3 / stride 2 max pool, TF 'SAME' on an even side: window rows 2oy .. 2oy+2 clipped to the image (padding never wins)
__device__ __forceinline__ void hg_pool(const float* src, int s_stride, int side_in, float* dst, int d_stride, int C) {
    const int so = side_in >> 1, c4n = C >> 2;
    for (int i = threadIdx.x; i < so * so * c4n; i += blockDim.x) {
        const int c4 = i % c4n, px = i / c4n, ox = px % so, oy = px / so;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy + ky;
            if (iy >= side_in) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox + kx;
                if (ix >= side_in) continue;
                const float4 v = *reinterpret_cast<const float4*>(src + (long)(iy * side_in + ix) * s_stride + c4 * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        *reinterpret_cast<float4*>(dst + px * d_stride + c4 * 4) = m;
    }
}

// dst[y][x] += lo[y/2][x/2]   (nearest-neighbour upsample + add, um_v1.py:66-69)
__device__ __forceinline__ void hg_upadd(float* dst, int side, const float* lo, int stride, int C) {
    const int c4n = C >> 2;
    for (int i = threadIdx.x; i < side * side * c4n; i += blockDim.x) {
        const int c4 = i % c4n, px = i / c4n, x = px % side, y = px / side;
        float4* d = reinterpret_cast<float4*>(dst + px * stride + c4 * 4);
        const float4 l = *reinterpret_cast<const float4*>(lo + ((y >> 1) * (side >> 1) + (x >> 1)) * stride + c4 * 4);
        float4 v = *d;
        v.x += l.x; v.y += l.y; v.z += l.z; v.w += l.w;
        *d = v;
    }
}

// One convolution + folded BatchReNorm + ReLU (+ in-place skip) on an LDS-resident crop.  RT4: 64 pixels = four 16-row tiles;
// otherwise one row tile (M = 16 or 4 live rows) and the four steps of a K-group accumulate into four independent tiles.
// src [M][Cin] (row stride s_stride) -> dst [M][Cout] (row stride d_stride); add_dst: dst += (the residual skip, in place);
// gout != null: the result goes to HBM (row stride g_stride) instead of dst.
template <bool RT4>
__device__ __forceinline__ void hg_conv(const float* src, int s_stride, int Cin, int side, int M, int ksize, const HgConvDesc& d, int Cout,
                                        float* dst, int d_stride, bool add_dst, float* gout, int g_stride, const float* zrow) {
    constexpr int NRT = RT4 ? 4 : 1;               // row tiles
    constexpr int QC = 2;                          // column tiles per wave (Cout <= 128)
    constexpr int GB = 4;                          // K-groups whose weight fragments are fetched as one batch, one batch ahead
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int CT = Cout >> 4;
    const int taps = ksize * ksize, pad = ksize >> 1;
    const int NG = (Cin >> 4) * taps;              // K-groups of 16: (16-channel chunk, tap), taps innermost -- the packing order
    const int side_shift = side == 8 ? 3 : (side == 4 ? 2 : 1);
    // this lane's weight rows: output channel n = 16 ct + r of column tiles ct = wave, wave + 4 (clamped: a dead tile recomputes the
    // wave's first one and is not stored)
    int wofs[QC];
    bool ct_ok[QC];
#pragma unroll
    for (int q = 0; q < QC; ++q) {
        const int ct = wave + 4 * q;
        ct_ok[q] = ct < CT;
        wofs[q] = ((ct_ok[q] ? ct : (wave < CT ? wave : 0)) * 16 + r) * 16 + 4 * g;
    }
    const bool wave_live = wave < CT;
    // this lane's pixel per row tile
    int py[NRT], px[NRT];
    bool p_ok[NRT];
#pragma unroll
    for (int t = 0; t < NRT; ++t) {
        const int p = t * 16 + r;
        p_ok[t] = p < M;
        py[t] = p >> side_shift; px[t] = p & (side - 1);
    }
    dr_f32x4 acc[4][QC];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < QC; ++q)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[a][q][v] = 0.f;

    float4 bn[GB][QC], bc[GB][QC];
    auto fetch = [&](int g0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < GB; ++j) {
            const int gi = g0 + j < NG ? g0 + j : NG - 1;           // (a group past the end re-reads the last one: dropped below)
#pragma unroll
            for (int q = 0; q < QC; ++q) bn[j][q] = *reinterpret_cast<const float4*>(d.w + (long)gi * d.Np * 16 + wofs[q]);
        }
    };
    if (wave_live) fetch(0);
    for (int g0 = 0; g0 < NG; g0 += GB) {
        if (!wave_live) break;
#pragma unroll
        for (int j = 0; j < GB; ++j)
#pragma unroll
            for (int q = 0; q < QC; ++q) bc[j][q] = bn[j][q];
        if (g0 + GB < NG) fetch(g0 + GB);
#pragma unroll
        for (int j = 0; j < GB; ++j) {
            const int gi = g0 + j;
            if (gi >= NG) break;                                     // wave-uniform
            const int chunk = gi / taps, tap = gi - chunk * taps;
            const int dy = tap / ksize - pad, dx = tap - (tap / ksize) * ksize - pad;
            float4 a[NRT];
#pragma unroll
            for (int t = 0; t < NRT; ++t) {
                const int yy = py[t] + dy, xx = px[t] + dx;
                const bool ok = p_ok[t] && yy >= 0 && yy < side && xx >= 0 && xx < side;
                const float* ap = ok ? src + ((yy << side_shift) + xx) * s_stride : zrow;
                a[t] = *reinterpret_cast<const float4*>(ap + chunk * 16 + 4 * g);
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int t = 0; t < NRT; ++t)
#pragma unroll
                    for (int q = 0; q < QC; ++q) {
                        const float av = s4 == 0 ? a[t].x : s4 == 1 ? a[t].y : s4 == 2 ? a[t].z : a[t].w;
                        const float bv = s4 == 0 ? bc[j][q].x : s4 == 1 ? bc[j][q].y : s4 == 2 ? bc[j][q].z : bc[j][q].w;
                        const int ai = RT4 ? t : s4;
                        acc[ai][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[ai][q], 0, 0, 0);
                    }
        }
    }
    // epilogue: accumulator register v of a tile is (row 4 g + v, column r)
#pragma unroll
    for (int q = 0; q < QC; ++q) {
        if (!ct_ok[q]) continue;
        const int n = (wave + 4 * q) * 16 + r;
        const float sc = d.scale[n], sh = d.shift[n];
#pragma unroll
        for (int t = 0; t < NRT; ++t) {
            dr_f32x4 sum = acc[RT4 ? t : 0][q];
            if (!RT4) sum = (sum + acc[1][q]) + (acc[2][q] + acc[3][q]);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int p = t * 16 + 4 * g + v;
                if (p >= M) continue;
                float val = fmaxf(sum[v] * sc + sh, 0.f);
                if (add_dst) val += dst[p * d_stride + n];
                if (gout) gout[(long)p * g_stride + n] = val;
                else dst[p * d_stride + n] = val;
            }
        }
    }
}

template <bool RT4>
__device__ __forceinline__ void hg_residual(float* buf, int sF, int F, int side, int M, const HgConvDesc* d, float* T1, float* T2, int sH,
                                            const float* zrow, float* gout, int g_stride) {
    hg_conv<RT4>(buf, sF, F, side, M, 1, d[0], F / 2, T1, sH, false, nullptr, 0, zrow);
    __syncthreads();
    hg_conv<RT4>(T1, sH, F / 2, side, M, 3, d[1], F / 2, T2, sH, false, nullptr, 0, zrow);
    __syncthreads();
    hg_conv<RT4>(T2, sH, F / 2, side, M, 1, d[2], F, buf, sF, true, gout, g_stride, zrow);
    __syncthreads();
}

__global__ __launch_bounds__(256, 1) void hg_tail_eval_kernel(const HgFusedParams p) {
    DR_DYN_SMEM(smem_raw);
    float* lds = reinterpret_cast<float*>(smem_raw);
    const int F = p.F, sF = F + 4, sH = F / 2 + 4;
    float* A = lds;
    float* T1 = A + 64 * sF;
    float* T2 = T1 + 64 * sH;
    float* C = T2 + 64 * sH;
    float* D = C + 16 * sF;
    float* Z = D + 4 * sF;
    for (int i = threadIdx.x; i < sF; i += blockDim.x) Z[i] = 0.f;
    const int b = blockIdx.x;
    hg_pool(p.x + (long)b * 256 * p.x_cs + p.x_coff, p.x_cs, 16, A, sF, F);
    __syncthreads();
    hg_residual<true>(A, sF, F, 8, 64, p.conv + 0, T1, T2, sH, Z, nullptr, 0);            // lower1 @8
    hg_pool(A, sF, 8, C, sF, F);
    __syncthreads();
    hg_residual<true>(A, sF, F, 8, 64, p.conv + 3, T1, T2, sH, Z, nullptr, 0);            // upper1 @8
    hg_residual<false>(C, sF, F, 4, 16, p.conv + 6, T1, T2, sH, Z, nullptr, 0);           // lower1 @4
    hg_pool(C, sF, 4, D, sF, F);
    __syncthreads();
    hg_residual<false>(C, sF, F, 4, 16, p.conv + 9, T1, T2, sH, Z, nullptr, 0);           // upper1 @4
    hg_residual<false>(D, sF, F, 2, 4, p.conv + 12, T1, T2, sH, Z, nullptr, 0);           // lower1 @2
    hg_residual<false>(D, sF, F, 2, 4, p.conv + 15, T1, T2, sH, Z, nullptr, 0);           // lower3 @2
    hg_upadd(C, 4, D, sF, F);
    __syncthreads();
    hg_residual<false>(C, sF, F, 4, 16, p.conv + 18, T1, T2, sH, Z, nullptr, 0);          // lower3 @4
    hg_upadd(A, 8, C, sF, F);
    __syncthreads();
    hg_residual<true>(A, sF, F, 8, 64, p.conv + 21, T1, T2, sH, Z, p.y + (long)b * 64 * p.y_cs + p.y_coff, p.y_cs);   // lower3 @8 -> HBM
}

}  // namespace dr
