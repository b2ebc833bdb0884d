// conv_wgrad_bf16.h -- weight gradient on the bf16 matrix cores (dr_set_precision(DR_PREC_BF16) on a training handle).
//
// dW[tap][ci][co] = sum_pix x[pix + tap][ci] * g[pix][co]: the contraction index is the PIXEL, so both MFMA operands
// need pixel-contiguous rows -- the transpose of how NHWC tensors lie in HBM.  v_mfma_f32_32x32x16_bf16 takes, per lane,
// 8 consecutive k of one row: lane (row li, half lk) reads 16 bytes = pixels 8*lk..8*lk+7 (+16 for the second k-step)
// of channel li.  The LDS image is therefore channel-major, Xt[ci][32 pixels] / Gt[co][32 pixels] in bf16: 64-byte rows of
// four 16-byte slots, swizzled like the forward kernel's tiles (slot ^ ((row >> 2) & 3): the 16 rows a ds_read_b128
// lane group touches land on 16 different bank positions).
// The transpose happens in registers while staging: a thread owns a unit (4 channels, 8 pixels) -- eight 16-byte global
// loads (coalesced: the 32 lanes of a pixel octet read 512 contiguous bytes per pixel), sixteen v_cvt_pk_bf16_f32, four
// 16-byte LDS stores (one per channel; 4-way bank conflict on the store side, accepted: rotating the channel order
// per lane would remove it at the price of ~50 VALU selects per step).  T = 128: 128 X units + 128 G units = one per
// thread.  Everything else -- the grid of (tile, tap, pixel slab) workgroups, the XCD-aware slab mapping, the partial
// slabs folded by wgrad_fold_all_kernel -- is the fp32 kernel's (conv_wgrad.h), so the planner is shared.
#pragma once
#include <type_traits>

#include "conv_wgrad.h"

namespace dr {

template <int T, int G16 = 0>      // G16: the g operand is stored as bf16 (WgradParams::g_bf16)
__global__ __launch_bounds__(256, (T == 128 ? 2 : 4)) void conv_wgrad_bf16_kernel(const WgradParams p) {
    constexpr int BKP = 32;                // pixels per step = two MFMA k-steps of 16
    constexpr int WT = T / 2;              // wave tile
    constexpr int TM = WT / 32;
    constexpr int C4N = T / 4;             // 4-channel groups per operand tile
    constexpr int UNITS = C4N * (BKP / 8); // (channel group, pixel octet) units per operand and step
    static_assert(2 * UNITS <= 256, "one unit per thread at most");
    __shared__ __attribute__((aligned(16))) float Xt[2][T][16];
    __shared__ __attribute__((aligned(16))) float Gt[2][T][16];

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

    // ---- loader: this thread's unit -----------------------------------------------------------------
    const int opnd = tid / UNITS;                                         // 0 = x, 1 = g, beyond = no unit
    const int un = tid % UNITS;
    const int c4 = (un % C4N) * 4, po = un / C4N;                         // 4 channels, pixel octet 0..3
    const bool is_x = opnd == 0, has_unit = opnd < 2;
    const int cbase = is_x ? ci0 + c4 : co0 + c4;
    const int cleft = (is_x ? p.Cin : p.Cout) - cbase;
    const int nvc = !has_unit ? 0 : (cleft < 0 ? 0 : (cleft > 4 ? 4 : cleft));   // valid channels of the float4
    const float* const src = is_x ? p.x : p.g;
    const unsigned cs = (unsigned)(is_x ? p.x_cs : p.g_cs);
    const unsigned coff = (unsigned)((is_x ? p.x_coff : p.g_coff) + cbase);
    const int tap_shift = is_x ? dy * p.W + dx : 0;
    const bool border = is_x && p.ksize > 1;
    const bool masked = is_x && p.rowmask != nullptr;
    const bool src16 = G16 && !is_x;                                      // wave-uniform: a wave stages either x or g units

    float4 v[8];
    unsigned okbits = 0;                                                  // bit q: pixel q of the octet is real data
    int next_step = 0;
    auto load = [&]() __attribute__((always_inline)) {
        const int mb = m_begin + next_step * BKP + po * 8;
        ++next_step;
        okbits = 0;
        float mk[8];
        unsigned eo[8];                                                   // element offset of (pixel, channel group) or 0
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int m = mb + q;
            bool ok = m < m_end && nvc > 0;
            if (border) {
                const int mm = ok ? m : 0;
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
            eo[q] = ok ? ms * cs + coff : 0u;
            if (masked) mk[q] = p.rowmask[ms];
            okbits |= (ok ? 1u : 0u) << q;
        }
        // the eight loads of a unit as ONE batch of the same instruction (a branch between them makes hipcc wait for each):
        // the wave-uniform "is this the bf16-stored g operand" decision is taken once, around the batch
        if (G16 && src16) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {                                 // 4 bf16 channels = 8 bytes, kept as bit patterns
                const float2 w2 = *reinterpret_cast<const float2*>(reinterpret_cast<const __bf16*>(src) + eo[q]);
                v[q] = make_float4(w2.x, w2.y, 0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const float4*>(src + eo[q]);
        }
        if (masked) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (mk[q] < p.mask_thresh) okbits &= ~(1u << q);
        }
    };
    auto store = [&](const int buf) __attribute__((always_inline)) {
        if (!has_unit) return;
        float (*dst)[16] = is_x ? Xt[buf] : Gt[buf];
        if (G16 && src16) {
            // already bf16: pixel q holds channels (0,1) in word x and (2,3) in word y; a channel's slot is its 16-bit half of
            // eight pixels.  Pixels that are not data are zeroed first (two selects each); channels beyond Cout need nothing:
            // a group of four is zero-padded by the producer and a group entirely beyond Cout never loads (nvc = 0).
            unsigned wx[8], wy[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bool live = (okbits >> q) & 1u;
                wx[q] = live ? __builtin_bit_cast(unsigned, v[q].x) : 0u;
                wy[q] = live ? __builtin_bit_cast(unsigned, v[q].y) : 0u;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned w[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const unsigned a0 = (j >> 1) ? wy[2 * h] : wx[2 * h], a1 = (j >> 1) ? wy[2 * h + 1] : wx[2 * h + 1];
                    w[h] = dr_pack_halves(a0, a1, j & 1);
                }
                const int row = c4 + j;
                *reinterpret_cast<float4*>(&dst[row][(po ^ ((row >> 2) & 3)) * 4]) =
                    make_float4(__builtin_bit_cast(float, w[0]), __builtin_bit_cast(float, w[1]), __builtin_bit_cast(float, w[2]), __builtin_bit_cast(float, w[3]));
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dr_f32x8 f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float e = j == 0 ? v[q].x : j == 1 ? v[q].y : j == 2 ? v[q].z : v[q].w;
                f[q] = (((okbits >> q) & 1u) && j < nvc) ? e : 0.f;
            }
            const int row = c4 + j;
            *reinterpret_cast<float4*>(&dst[row][(po ^ ((row >> 2) & 3)) * 4]) = __builtin_bit_cast(float4, __builtin_convertvector(f, dr_bf16x8));
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
    const int sw = (li >> 2) & 3;                                         // rows wm*WT + 32*i + li: (row >> 2) & 3
    auto k_step = [&](const int buf, const bool more) __attribute__((always_inline)) {
        if (more) load();
        float4 a[2][TM], b[2][TM];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                a[g][t] = *reinterpret_cast<const float4*>(&Xt[buf][wm * WT + 32 * t + li][((g * 2 + lk) ^ sw) * 4]);
                b[g][t] = *reinterpret_cast<const float4*>(&Gt[buf][wn * WT + 32 * t + li][((g * 2 + lk) ^ sw) * 4]);
            }
        auto mf = [&](auto NA, auto NB) __attribute__((always_inline)) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int i = 0; i < decltype(NA)::value; ++i)
#pragma unroll
                    for (int j = 0; j < decltype(NB)::value; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dr_bf16x8, a[g][i]),
                                                                            __builtin_bit_cast(dr_bf16x8, b[g][j]), acc[i][j], 0, 0, 0);
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
}

}  // namespace dr
