// conv_wgrad_tr.h -- weight gradient on the bf16 matrix cores with the operand transpose done by the LDS hardware.
//
// dW[tap][ci][co] = sum_pix x[pix + tap][ci] * g[pix][co]: the contraction index is the PIXEL, so v_mfma_f32_32x32x16_bf16 wants,
// per lane, 8 consecutive pixels of one channel -- the transpose of NHWC.  conv_wgrad_bf16.h transposes in registers while
// staging (a thread owns 4 channels x 8 pixels: 16 conversions, 32 selects, four conflicting LDS stores per unit) and that
// staging, not the matrix cores or HBM, bounds the kernel (141 TFLOP/s of 2500; storing g as bf16 -- half the bytes -- made it
// no faster).  Here the tiles lie in LDS as they lie in HBM, [pixel][channel] in bf16: staging is one conversion pass and a
// 16-byte store per 8 channels (or a straight copy when g is stored as bf16), and each fragment is two
// ds_read_b64_tr_b16 (gfx950): a group of 16 lanes reads a [4 pixels][16 channels] block and every lane receives the four
// pixels of ITS channel.  Rows are padded by 16 channels (288- / 160-byte stride) so the four rows of a block fall on
// different banks.  Grid, slab planner and fold are conv_wgrad.h's.
#pragma once
#include <type_traits>

#include "conv_wgrad.h"

namespace dr {

template <int T, int G16, int X16 = 0>      // G16 / X16: the g / x operand is stored as bf16
__global__ __launch_bounds__(256, (T == 128 ? 2 : 4)) void conv_wgrad_tr_kernel(const WgradParams p) {
    constexpr int BKP = 32;                // pixels per step = two MFMA k-steps of 16
    constexpr int WT = T / 2;              // wave tile
    constexpr int TM = WT / 32;
    constexpr int RS = T + 16;             // LDS row stride in bf16 elements (pad: the 4 rows of a transpose block on 4 bank groups)
    constexpr int C8N = T / 8;             // 8-channel chunks per pixel row
    constexpr int CHUNKS = BKP * C8N;      // chunks per operand and step
    constexpr int IT = CHUNKS / 256;       // chunks per thread and operand (2 for T = 128, 1 for T = 64)
    static_assert(CHUNKS % 256 == 0, "chunk mapping");
    __shared__ __attribute__((aligned(16))) unsigned short Xs0[BKP][RS];
    __shared__ __attribute__((aligned(16))) unsigned short Xs1[BKP][RS];
    __shared__ __attribute__((aligned(16))) unsigned short Gs0[BKP][RS];
    __shared__ __attribute__((aligned(16))) unsigned short Gs1[BKP][RS];
#define DR_XS(st) ((st) ? Xs1 : Xs0)
#define DR_GS(st) ((st) ? Gs1 : Gs0)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lk = lane >> 5, li = lane & 31;
    const int co_tiles = dr_ceil_div(p.Cout, T);
    const int taps = p.ksize * p.ksize;
    const int tiles = dr_ceil_div(p.Cin, T) * co_tiles;
    int split, rest;
    if ((p.nsplit & 7) == 0) {                                            // slab s on XCD s % 8 (conv_wgrad_kernel)
        const int per = p.nsplit >> 3, j = blockIdx.x >> 3;
        split = (j % per) * 8 + (blockIdx.x & 7);
        rest = j / per;
    } else {
        split = blockIdx.x % p.nsplit;
        rest = blockIdx.x / p.nsplit;
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
    const int steps = m_begin < m_end ? (m_end - m_begin + BKP - 1) / BKP : 0;
    const bool pow2 = (p.W & (p.W - 1)) == 0 && (HW & (HW - 1)) == 0;
    const int w_shift = __builtin_ctz((unsigned)p.W);
    const int tap_shift = dy * p.W + dx;
    const bool border = p.ksize > 1;

    // ---- staging: chunk = 8 channels of one pixel ---------------------------------------------------------------------
    int c_pix[IT], c_ch[IT];                          // pixel row of the step, first channel of the chunk (within the tile)
    int x_nv[IT], g_nv[IT];                           // valid channels of the chunk (0..8)
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int id = tid + i * 256;
        c_pix[i] = id / C8N;
        c_ch[i] = (id % C8N) * 8;
        const int xl = p.Cin - (ci0 + c_ch[i]), gl = p.Cout - (co0 + c_ch[i]);
        x_nv[i] = xl < 0 ? 0 : (xl > 8 ? 8 : xl);
        g_nv[i] = gl < 0 ? 0 : (gl > 8 ? 8 : gl);
    }
    float4 xa[IT], xb[IT], ga[IT], gb[IT];            // fp32: channels 0..3 / 4..7 of the chunk; bf16-stored g: ga = the chunk
    unsigned x_ok = 0, g_ok = 0;                      // bit i: chunk i of this step holds data
    int next_step = 0;
    auto load = [&]() __attribute__((always_inline)) {
        const int mb = m_begin + next_step * BKP;
        ++next_step;
        x_ok = g_ok = 0;
        unsigned xo[IT], go[IT];
        float mk[IT];
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int m = mb + c_pix[i];
            const bool in = m < m_end;
            bool okx = in && x_nv[i] > 0;
            if (border) {
                const int mm = in ? m : 0;
                int yy, xx;
                if (pow2) { const int rem = mm & (HW - 1); yy = (rem >> w_shift) + dy; xx = (rem & (p.W - 1)) + dx; }
                else { const int rem = mm % HW; yy = rem / p.W + dy; xx = rem % p.W + dx; }
                okx = okx && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            }
            const unsigned ms = okx ? (unsigned)(m + tap_shift) : 0u;
            xo[i] = okx ? ms * (unsigned)p.x_cs + (unsigned)(p.x_coff + ci0 + c_ch[i]) : 0u;
            if (p.rowmask) mk[i] = p.rowmask[ms];
            const bool okg = in && g_nv[i] > 0;
            go[i] = okg ? (unsigned)m * (unsigned)p.g_cs + (unsigned)(p.g_coff + co0 + c_ch[i]) : 0u;
            x_ok |= (okx ? 1u : 0u) << i;
            g_ok |= (okg ? 1u : 0u) << i;
        }
        // one batch of unconditional loads (a dead chunk reads the tensor base and is zeroed at store time); the second half of
        // an fp32 chunk is only fetched where the row has it (a chunk of <= 4 valid channels ends at the row's last 16 bytes)
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            if constexpr (X16) {
                xa[i] = dr_load16_a4(reinterpret_cast<const __bf16*>(p.x) + xo[i]);
            } else {
                xa[i] = *reinterpret_cast<const float4*>(p.x + xo[i]);
                xb[i] = *reinterpret_cast<const float4*>(p.x + (x_nv[i] > 4 ? xo[i] + 4u : xo[i]));
            }
            if constexpr (G16) {
                ga[i] = dr_load16_a4(reinterpret_cast<const __bf16*>(p.g) + go[i]);
            } else {
                ga[i] = *reinterpret_cast<const float4*>(p.g + go[i]);
                gb[i] = *reinterpret_cast<const float4*>(p.g + (g_nv[i] > 4 ? go[i] + 4u : go[i]));
            }
        }
        if (p.rowmask) {
#pragma unroll
            for (int i = 0; i < IT; ++i)
                if (mk[i] < p.mask_thresh) x_ok &= ~(1u << i);
        }
    };
    // fp32 chunk -> 8 bf16 (nearest even), channels beyond nv and dead chunks zero
    auto pack = [](float4 a, float4 b, int nv, bool live) -> float4 {
        const int n = live ? nv : 0;
        const dr_f32x8 f = {n > 0 ? a.x : 0.f, n > 1 ? a.y : 0.f, n > 2 ? a.z : 0.f, n > 3 ? a.w : 0.f,
                            n > 4 ? b.x : 0.f, n > 5 ? b.y : 0.f, n > 6 ? b.z : 0.f, n > 7 ? b.w : 0.f};
        return __builtin_bit_cast(float4, __builtin_convertvector(f, dr_bf16x8));
    };
    auto store = [&](const int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            if constexpr (X16) {
                float4 w16 = xa[i];
                const bool live = (x_ok >> i) & 1u;
                if (!live) w16 = make_float4(0.f, 0.f, 0.f, 0.f);
                else if (x_nv[i] <= 4) { w16.z = 0.f; w16.w = 0.f; }
                *reinterpret_cast<float4*>(&DR_XS(buf)[c_pix[i]][c_ch[i]]) = w16;
            } else {
                *reinterpret_cast<float4*>(&DR_XS(buf)[c_pix[i]][c_ch[i]]) = pack(xa[i], xb[i], x_nv[i], (x_ok >> i) & 1u);
            }
            if constexpr (G16) {
                // stored bf16: groups of four channels are zero-padded by the producer; a chunk whose upper half hangs over the
                // row (nv <= 4) or that is dead needs masking
                float4 w16 = ga[i];
                const bool live = (g_ok >> i) & 1u;
                if (!live) w16 = make_float4(0.f, 0.f, 0.f, 0.f);
                else if (g_nv[i] <= 4) { w16.z = 0.f; w16.w = 0.f; }
                *reinterpret_cast<float4*>(&DR_GS(buf)[c_pix[i]][c_ch[i]]) = w16;
            } else {
                *reinterpret_cast<float4*>(&DR_GS(buf)[c_pix[i]][c_ch[i]]) = pack(ga[i], gb[i], g_nv[i], (g_ok >> i) & 1u);
            }
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
    const int na_ = (p.Cin - (ci0 + wm * WT) + 31) / 32, nb_ = (p.Cout - (co0 + wn * WT) + 31) / 32;
    const int na = na_ < 0 ? 0 : (na_ > TM ? TM : na_), nb = nb_ < 0 ? 0 : (nb_ > TM ? TM : nb_);
    // transpose-read addressing: lane = 16 * grp + i16; grp & 1 selects the 16-channel half of the 32-channel MFMA tile, grp >> 1 = lk
    // the k half; the lane SUPPLIES chunk i16 of the [4 pixels][16 channels] block: pixel i16 / 4, channels 4 * (i16 % 4)
    const int i16 = lane & 15, half = (lane >> 4) & 1;
    const int t_pix = 8 * lk + (i16 >> 2);                                // + 16 * kstep + 4 * r
    const int t_ch = 16 * half + 4 * (i16 & 3);                          // + tile base
    auto k_step = [&](const int buf, const bool more) __attribute__((always_inline)) {
        if (more) load();
        dr_bf16x8 a[2][TM], b[2][TM];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const uint2 a0 = dr_lds_read_tr16(&DR_XS(buf)[16 * ks + t_pix][wm * WT + 32 * t + t_ch]);
                const uint2 a1 = dr_lds_read_tr16(&DR_XS(buf)[16 * ks + t_pix + 4][wm * WT + 32 * t + t_ch]);
                const uint2 b0 = dr_lds_read_tr16(&DR_GS(buf)[16 * ks + t_pix][wn * WT + 32 * t + t_ch]);
                const uint2 b1 = dr_lds_read_tr16(&DR_GS(buf)[16 * ks + t_pix + 4][wn * WT + 32 * t + t_ch]);
                a[ks][t] = __builtin_bit_cast(dr_bf16x8, make_float4(__builtin_bit_cast(float, a0.x), __builtin_bit_cast(float, a0.y),
                                                                      __builtin_bit_cast(float, a1.x), __builtin_bit_cast(float, a1.y)));
                b[ks][t] = __builtin_bit_cast(dr_bf16x8, make_float4(__builtin_bit_cast(float, b0.x), __builtin_bit_cast(float, b0.y),
                                                                      __builtin_bit_cast(float, b1.x), __builtin_bit_cast(float, b1.y)));
            }
        auto mf = [&](auto NA, auto NB) __attribute__((always_inline)) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < decltype(NA)::value; ++i)
#pragma unroll
                    for (int j = 0; j < decltype(NB)::value; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
        };
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        if constexpr (TM == 2) {
            if (na == 2) {
                if (nb == 2) mf(I2{}, I2{});
                else if (nb == 1) mf(I2{}, I1{});
            } else if (na == 1) {
                if (nb == 2) mf(I1{}, I2{});
                else if (nb == 1) mf(I1{}, I1{});
            }
        } else {
            if (na > 0 && nb > 0) mf(I1{}, I1{});
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

    // partial[split][tap][ci][co]; D: row = (r&3)+8*(r>>2)+4*lk (ci), col = li (co)
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
#undef DR_XS
#undef DR_GS
}

}  // namespace dr
