// conv_wgrad16.h -- weight gradient of the NARROW layers on 16x16 matrix-core tiles (v_mfma_f32_16x16x4_f32).
//
//   dW[tap][ci][co] = sum over pixels m of  x[m shifted by tap][ci] * g[m][co]          (conv_wgrad.h; same slab layout)
//
// The square kernels of conv_wgrad.h compute 32x32-channel MFMA tiles inside 64- or 128-channel blocks; a layer with 16, 32, 65 or
// 78 channels on a side pays for the padding in matrix-core time: 3x3 78->78 on the 96-channel row kernel is 66 % live (65->65:
// 46 %), 1x1 156->78 on 64-channel blocks 50 %, the stem's 3x3 16->16 6 %.  The 16x16x4 instruction has the same rate per flop and
// the same exact fp32 arithmetic per product as 32x32x2 (conv_igemm.h, MF = 16, uses it for the forward's narrow outputs), with
// output tiles of 16 x 16 channels: 78 channels are 5 tiles (80 computed), 65 are 5, 156 are 10, 16 are 1.
//
// A workgroup owns, for one pixel slab: ND taps (ND = 3: the three taps dx = -1, 0, +1 of one kernel row dy, which share the G
// tile like conv_wgrad_row_kernel; ND = 1: one tap) x NI x 16 input channels x NJ x 16 output channels -- ND * NI row blocks of
// NJ tiles each.  The four waves take row blocks w, w + 4, ... (COLS = 0) or tile columns w, w + 4, ... (COLS = 1: few rows, many
// columns), every tile of a row block / column; a wave with one block less than the others computes a clamped duplicate that is
// not stored (no branch in the loop: a run-time "tile is live" test serialises every LDS read behind its own wait, conv_wgrad.h).
// Lane (r = lane & 15, g = lane >> 4) feeds MFMA step s of a 16-pixel tile with A = x[pixel 4s + g][16 i + r] and
// B = g[pixel 4s + g][16 j + r]: both straight out of the pixel-major LDS image the loads produce (no transpose); the row stride is
// 16 (mod 32) floats, so the four pixel rows of a read fall on the two halves of the banks.  Accumulator register v of a tile is
// dW[16 i + 4 g + v][16 j + r].  Slabs, fold and determinism as in conv_wgrad.h: partial[split][tap][Cin][Cout], fixed-order fold.
#pragma once
#include "conv_wgrad.h"

namespace dr {

template <int ND, int NI, int NJ, int COLS>
struct Wgrad16Cfg {
    static constexpr int BK = 16;                                  // pixels per LDS tile
    static constexpr int XW = NI * 16, GW = NJ * 16;               // channels staged per pixel row
    static constexpr int XST = XW + (XW % 32 == 16 ? 0 : 16), GST = GW + (GW % 32 == 16 ? 0 : 16);   // row strides: 16 mod 32
    static constexpr int XC4 = XW / 4, GC4 = GW / 4;               // float4 per pixel row
    static constexpr int XI = (ND * BK * XC4 + 255) / 256, GI = (BK * GC4 + 255) / 256;   // float4 loads per thread and tile
    static constexpr int NRB = ND * NI;                            // row blocks (tap, 16 input channels)
    static constexpr int QR = COLS ? NRB : (NRB + 3) / 4;          // row blocks per wave
    static constexpr int QC = COLS ? (NJ + 3) / 4 : NJ;            // tile columns per wave
    // resident waves per SIMD asked of the register allocator: 4 QR QC accumulator registers + ~40 of staged loads; at three waves
    // (168 registers) the 15- and 20-tile variants spill 64 / 192 bytes per lane into the pixel loop
    static constexpr int kWaves = QR * QC >= 15 ? 2 : 4;
};

// workgroups per pixel slab: (kernel rows or taps) x ci blocks x co blocks
template <int ND, int NI, int NJ>
__host__ __device__ inline int wgrad16_blocks_per_slab(int Cin, int Cout, int ksize) {
    const int tapg = ksize == 3 ? (ND == 3 ? 3 : 9) : 1;
    return tapg * dr_ceil_div(Cin, NI * 16) * dr_ceil_div(Cout, NJ * 16);
}

template <int ND, int NI, int NJ, int COLS>
__global__ __launch_bounds__(256, (Wgrad16Cfg<ND, NI, NJ, COLS>::kWaves)) void conv_wgrad16_kernel(const WgradParams p) {
    using K = Wgrad16Cfg<ND, NI, NJ, COLS>;
    constexpr int BK = K::BK;
    static_assert(ND == 1 || ND == 3, "one tap, or the three taps of a kernel row");
    __shared__ __attribute__((aligned(16))) float Xs[2][ND][BK][K::XST];
    __shared__ __attribute__((aligned(16))) float Gs[2][BK][K::GST];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g4 = lane >> 4;
    // grid = blocks-per-slab x nsplit; slab s on XCD s % 8 when nsplit % 8 == 0 (conv_wgrad.h)
    const int ci_blocks = dr_ceil_div(p.Cin, K::XW), co_blocks = dr_ceil_div(p.Cout, K::GW);
    int split, rest;
    if ((p.nsplit & 7) == 0) {
        const int per = p.nsplit >> 3, j = blockIdx.x >> 3;
        split = (j % per) * 8 + (blockIdx.x & 7);
        rest = j / per;
    } else {
        split = blockIdx.x % p.nsplit;
        rest = blockIdx.x / p.nsplit;
    }
    const int cob = rest % co_blocks, cib = (rest / co_blocks) % ci_blocks, tapg = rest / (co_blocks * ci_blocks);
    const int ci0 = cib * K::XW, co0 = cob * K::GW;
    // ND = 3: tapg = kernel row (dy + 1), the workgroup's taps are tapg * 3 + {0, 1, 2}; ND = 1: tapg = the tap
    const int pad = p.ksize / 2;
    const int dy = (ND == 3 ? tapg : tapg / p.ksize) - pad;
    const int dx0 = (ND == 3 ? 0 : tapg % p.ksize) - pad;          // dx of the workgroup's first tap
    const int HW = p.H * p.W;
    const int M = p.B * HW;
    const int m_begin = split * p.rows_per_split;
    const int m_end = m_begin + p.rows_per_split < M ? m_begin + p.rows_per_split : M;
    const int steps = m_begin < m_end ? (m_end - m_begin + BK - 1) / BK : 0;
    const bool pow2 = (p.W & (p.W - 1)) == 0 && (HW & (HW - 1)) == 0;
    const int w_shift = __builtin_ctz((unsigned)p.W);

    // ---- loader (the mapping of conv_wgrad_row_kernel with compile-time tile widths) -------------------------------------------
    float4 xr[K::XI], gr[K::GI];
    float xm[K::XI];
    int xnv[K::XI], gnv[K::GI];
    int next_step = 0;
    auto load = [&]() __attribute__((always_inline)) {
        const int mb = m_begin + next_step * BK;
        ++next_step;
#pragma unroll
        for (int i = 0; i < K::XI; ++i) {
            const int idx = tid + i * 256;
            const int d = idx / (BK * K::XC4), rem = idx % (BK * K::XC4);
            const int row = rem / K::XC4, c4 = (rem % K::XC4) * 4;
            const int m = mb + row;
            const int left = p.Cin - (ci0 + c4);
            const int nv = left < 0 ? 0 : (left > 4 ? 4 : left);
            bool ok = idx < ND * BK * K::XC4 && m < m_end && nv > 0;
            unsigned ms = (unsigned)(ok ? m : 0);
            if (p.ksize > 1) {
                const int mm = ok ? m : 0;
                int y, x;
                if (pow2) { const int r = mm & (HW - 1); y = r >> w_shift; x = r & (p.W - 1); }
                else { const int r = mm % HW; y = r / p.W; x = r % p.W; }
                const int yy = y + dy, xx = x + dx0 + d;
                ok = ok && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                ms = ok ? (unsigned)(m + dy * p.W + dx0 + d) : 0u;
            }
            xr[i] = *reinterpret_cast<const float4*>(ok ? p.x + (ms * (unsigned)p.x_cs + (unsigned)(p.x_coff + ci0 + c4)) : p.x);
            if (p.rowmask) xm[i] = p.rowmask[ms];
            xnv[i] = ok ? nv : 0;
        }
#pragma unroll
        for (int i = 0; i < K::GI; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / K::GC4, c4 = (idx % K::GC4) * 4;
            const int m = mb + row;
            const int left = p.Cout - (co0 + c4);
            const int nv = left < 0 ? 0 : (left > 4 ? 4 : left);
            const bool ok = idx < BK * K::GC4 && m < m_end && nv > 0;
            gr[i] = *reinterpret_cast<const float4*>(ok ? p.g + ((unsigned)m * (unsigned)p.g_cs + (unsigned)(p.g_coff + co0 + c4)) : p.g);
            gnv[i] = ok ? nv : 0;
        }
    };
    auto zsel = [](float4 v, int nv) {
        return make_float4(nv > 0 ? v.x : 0.f, nv > 1 ? v.y : 0.f, nv > 2 ? v.z : 0.f, nv > 3 ? v.w : 0.f);
    };
    auto store = [&](const int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < K::XI; ++i) {
            const int idx = tid + i * 256;
            if (idx < ND * BK * K::XC4) {
                const int d = idx / (BK * K::XC4), rem = idx % (BK * K::XC4);
                const int nv = (p.rowmask && xm[i] < p.mask_thresh) ? 0 : xnv[i];
                *reinterpret_cast<float4*>(&Xs[buf][d][rem / K::XC4][(rem % K::XC4) * 4]) = zsel(xr[i], nv);
            }
        }
#pragma unroll
        for (int i = 0; i < K::GI; ++i) {
            const int idx = tid + i * 256;
            if (idx < BK * K::GC4) *reinterpret_cast<float4*>(&Gs[buf][idx / K::GC4][(idx % K::GC4) * 4]) = zsel(gr[i], gnv[i]);
        }
    };

    // ---- this wave's tiles: row blocks rb[q] (clamped: a duplicate is computed and dropped), columns cj[q] ------------------------
    int a_off[K::QR];                  // float offset of row block q inside one Xs stage: [d][.][16 i]
    int b_off[K::QC];                  // float offset of tile column q inside a Gs row
#pragma unroll
    for (int q = 0; q < K::QR; ++q) {
        int rb = COLS ? q : wave + 4 * q;
        rb = rb < K::NRB ? rb : K::NRB - 1;
        a_off[q] = (rb / NI) * BK * K::XST + (rb % NI) * 16 + r16;
    }
#pragma unroll
    for (int q = 0; q < K::QC; ++q) {
        int cj = COLS ? wave + 4 * q : q;
        cj = cj < NJ ? cj : NJ - 1;
        b_off[q] = cj * 16 + r16;
    }
    dr_f32x4 acc[K::QR][K::QC];
#pragma unroll
    for (int a = 0; a < K::QR; ++a)
#pragma unroll
        for (int b = 0; b < K::QC; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[a][b][v] = 0.f;

    if (steps > 0) {
        load();
        store(0);
    }
    __syncthreads();
    auto k_step = [&](const int buf, const bool more) __attribute__((always_inline)) {
        if (more) load();
        const float* xs = &Xs[buf][0][0][0];
        const float* gs = &Gs[buf][0][0];
#pragma unroll
        for (int s = 0; s < BK / 4; ++s) {
            float a[K::QR], b[K::QC];                            // every fragment of the k-step, read up front
#pragma unroll
            for (int q = 0; q < K::QR; ++q) a[q] = xs[(4 * s + g4) * K::XST + a_off[q]];
#pragma unroll
            for (int q = 0; q < K::QC; ++q) b[q] = gs[(4 * s + g4) * K::GST + b_off[q]];
#pragma unroll
            for (int qa = 0; qa < K::QR; ++qa)
#pragma unroll
                for (int qb = 0; qb < K::QC; ++qb)
                    acc[qa][qb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[qa], b[qb], acc[qa][qb], 0, 0, 0);
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

    // partial[split][tap][ci][co]; accumulator v of a tile = dW[16 i + 4 g + v][16 j + r]
    const int taps = p.ksize * p.ksize;
#pragma unroll
    for (int qa = 0; qa < K::QR; ++qa) {
        const int rb = COLS ? qa : wave + 4 * qa;
        if (rb >= K::NRB) continue;                              // the clamped duplicate
        const int d = rb / NI, i = rb % NI;
        const int tap = ND == 3 ? tapg * 3 + d : tapg;
        float* dst = p.partial + ((long)split * taps + tap) * p.Cin * p.Cout;
#pragma unroll
        for (int qb = 0; qb < K::QC; ++qb) {
            const int cj = COLS ? wave + 4 * qb : qb;
            if (cj >= NJ) continue;
            const int co = co0 + cj * 16 + r16;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int ci = ci0 + i * 16 + 4 * g4 + v;
                if (ci < p.Cin && co < p.Cout) dst[(long)ci * p.Cout + co] = acc[qa][qb][v];
            }
        }
    }
}

// The instantiations and when each is used (host side: train_exec.inc::wgrad_plan).  id -> (ND, NI, NJ, COLS):
//   1: (3, 5, 5, 0)   3x3, both sides 65..80 channels (the hm3 / um-head residuals: 65 -> 65, 78 -> 78)       one kernel row per workgroup
//   2: (3, 2, 2, 0)   3x3, both sides <= 32 channels (the stem's 16 -> 16, 32 -> 32)
//   3: (1, 5, 8, 1)   1x1, <= 80 input channels, output channels in blocks of 128 (78 -> 256, 65 -> 128, 70 -> 128)
//   4: (1, 10, 5, 0)  1x1, input channels in blocks of 160, <= 80 output channels (156 -> 78, 131 -> 65)
//   5: (1, 4, 4, 0)   1x1, both sides in blocks of 64 where the 64-channel MFMA block would be mostly padding (16 -> 64, 32 -> 16, ...)
enum { WG16_NONE = 0, WG16_ROW80 = 1, WG16_ROW32 = 2, WG16_IN80 = 3, WG16_OUT80 = 4, WG16_SMALL = 5 };

inline int wgrad16_blocks(int id, int Cin, int Cout, int ksize) {
    switch (id) {
        case WG16_ROW80: return wgrad16_blocks_per_slab<3, 5, 5>(Cin, Cout, ksize);
        case WG16_ROW32: return wgrad16_blocks_per_slab<3, 2, 2>(Cin, Cout, ksize);
        case WG16_IN80: return wgrad16_blocks_per_slab<1, 5, 8>(Cin, Cout, ksize);
        case WG16_OUT80: return wgrad16_blocks_per_slab<1, 10, 5>(Cin, Cout, ksize);
        case WG16_SMALL: return wgrad16_blocks_per_slab<1, 4, 4>(Cin, Cout, ksize);
        default: return 0;
    }
}

inline void launch_wgrad16(int id, const WgradParams& p, int grid, hipStream_t s) {
    switch (id) {
        case WG16_ROW80: DR_LAUNCH((conv_wgrad16_kernel<3, 5, 5, 0>), dim3(grid), dim3(256), 0, s, p); break;
        case WG16_ROW32: DR_LAUNCH((conv_wgrad16_kernel<3, 2, 2, 0>), dim3(grid), dim3(256), 0, s, p); break;
        case WG16_IN80: DR_LAUNCH((conv_wgrad16_kernel<1, 5, 8, 1>), dim3(grid), dim3(256), 0, s, p); break;
        case WG16_OUT80: DR_LAUNCH((conv_wgrad16_kernel<1, 10, 5, 0>), dim3(grid), dim3(256), 0, s, p); break;
        case WG16_SMALL: DR_LAUNCH((conv_wgrad16_kernel<1, 4, 4, 0>), dim3(grid), dim3(256), 0, s, p); break;
        default: break;
    }
}

}  // namespace dr
