// conv_wgrad_x3.h -- weight gradient with fp32-accurate products on the bf16 matrix cores (the operand split of conv_x3.h).
//
// dW[tap][ci][co] = sum_pix x[pix + tap][ci] * g[pix][co] on fp32 tensors (TF's Conv2DBackpropFilter of ops.py:282): both operands
// are split into three bf16 planes WHILE STAGED (x = x0 + x1 + x2 to 2^-24, round to nearest even), six of the nine plane products
// are formed on v_mfma_f32_32x32x16_bf16 -- the leading one into `acc`, the five corrections into their own accumulator (conv_x3.h:
// the same error class as the fp32 matrix cores, measured there) -- and the transpose the contraction over PIXELS needs is done by
// the LDS hardware (ds_read_b64_tr_b16), exactly as in conv_wgrad_tr.h, whose tile layout, grid, slab planner and fold this kernel
// shares.  A step is 16 pixels (one MFMA k-step per plane pair): three planes of two operands in two stages are 55 KB of LDS at
// T = 128 (two workgroups per CU), 31 KB at T = 64.
#pragma once
#include <type_traits>

#include "conv_wgrad.h"
#include "conv_x3.h"

namespace dr {

template <int T>
__global__ __launch_bounds__(256, (T == 128 ? 2 : 4)) void conv_wgrad_x3_kernel(const WgradParams p) {
    constexpr int BKP = 16;                // pixels per step
    constexpr int WT = T / 2;              // wave tile
    constexpr int TM = WT / 32;
    constexpr int RS = T + 16;             // LDS row stride in bf16 elements (conv_wgrad_tr.h: the 4 rows of a transpose block on 4 bank groups)
    constexpr int C8N = T / 8;             // 8-channel chunks per pixel row
    constexpr int CHUNKS = BKP * C8N;      // chunks per operand and step: 256 (T = 128) or 128 (T = 64: the upper half of the threads stage nothing)
    static_assert(CHUNKS == 256 || CHUNKS == 128, "chunk mapping");
    __shared__ __attribute__((aligned(16))) unsigned short Xs0[3][BKP][RS];
    __shared__ __attribute__((aligned(16))) unsigned short Xs1[3][BKP][RS];
    __shared__ __attribute__((aligned(16))) unsigned short Gs0[3][BKP][RS];
    __shared__ __attribute__((aligned(16))) unsigned short Gs1[3][BKP][RS];
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

    // ---- staging: chunk = 8 channels of one pixel, one per thread and operand ----------------------------------------------------
    const bool stager = tid < CHUNKS;
    const int c_pix = (tid % CHUNKS) / C8N, c_ch = ((tid % CHUNKS) % C8N) * 8;
    const int xl = p.Cin - (ci0 + c_ch), gl = p.Cout - (co0 + c_ch);
    const int x_nv = !stager ? 0 : xl < 0 ? 0 : (xl > 8 ? 8 : xl);        // valid channels of the chunk (0..8)
    const int g_nv = !stager ? 0 : gl < 0 ? 0 : (gl > 8 ? 8 : gl);
    float4 xa, xb, ga, gb;                                                // channels 0..3 / 4..7 of the chunk
    bool x_ok = false, g_ok = false;
    int next_step = 0;
    auto load = [&]() __attribute__((always_inline)) {
        const int m = m_begin + next_step * BKP + c_pix;
        ++next_step;
        const bool in = m < m_end;
        bool okx = in && x_nv > 0;
        if (border) {
            const int mm = in ? m : 0;
            int yy, xx;
            if (pow2) { const int rem = mm & (HW - 1); yy = (rem >> w_shift) + dy; xx = (rem & (p.W - 1)) + dx; }
            else { const int rem = mm % HW; yy = rem / p.W + dy; xx = rem % p.W + dx; }
            okx = okx && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
        }
        const unsigned ms = okx ? (unsigned)(m + tap_shift) : 0u;
        const unsigned xo = okx ? ms * (unsigned)p.x_cs + (unsigned)(p.x_coff + ci0 + c_ch) : 0u;
        float mk = 0.f;
        if (p.rowmask) mk = p.rowmask[ms];
        const bool okg = in && g_nv > 0;
        const unsigned go = okg ? (unsigned)m * (unsigned)p.g_cs + (unsigned)(p.g_coff + co0 + c_ch) : 0u;
        // one batch of unconditional loads (a dead chunk reads the tensor base and is zeroed at store time); the second half of a
        // chunk is only fetched where the row has it (a chunk of <= 4 valid channels ends at the row's last 16 bytes)
        xa = *reinterpret_cast<const float4*>(p.x + xo);
        xb = *reinterpret_cast<const float4*>(p.x + (x_nv > 4 ? xo + 4u : xo));
        ga = *reinterpret_cast<const float4*>(p.g + go);
        gb = *reinterpret_cast<const float4*>(p.g + (g_nv > 4 ? go + 4u : go));
        x_ok = okx && !(p.rowmask && mk < p.mask_thresh);
        g_ok = okg;
    };
    // fp32 chunk -> three planes of 8 bf16; channels beyond nv and dead chunks are zero
    auto split8 = [](float4 a, float4 b, int nv, bool live, float4& h0, float4& h1, float4& h2) __attribute__((always_inline)) {
        const int n = live ? nv : 0;
        const float4 lo4 = make_float4(n > 0 ? a.x : 0.f, n > 1 ? a.y : 0.f, n > 2 ? a.z : 0.f, n > 3 ? a.w : 0.f);
        const float4 hi4 = make_float4(n > 4 ? b.x : 0.f, n > 5 ? b.y : 0.f, n > 6 ? b.z : 0.f, n > 7 ? b.w : 0.f);
        uint2 l0, l1, l2, u0, u1, u2;
        x3_split4(lo4, l0, l1, l2);
        x3_split4(hi4, u0, u1, u2);
        h0 = make_float4(__builtin_bit_cast(float, l0.x), __builtin_bit_cast(float, l0.y), __builtin_bit_cast(float, u0.x), __builtin_bit_cast(float, u0.y));
        h1 = make_float4(__builtin_bit_cast(float, l1.x), __builtin_bit_cast(float, l1.y), __builtin_bit_cast(float, u1.x), __builtin_bit_cast(float, u1.y));
        h2 = make_float4(__builtin_bit_cast(float, l2.x), __builtin_bit_cast(float, l2.y), __builtin_bit_cast(float, u2.x), __builtin_bit_cast(float, u2.y));
    };
    auto store = [&](const int buf) __attribute__((always_inline)) {
        if (CHUNKS < 256 && !stager) return;
        float4 h0, h1, h2;
        split8(xa, xb, x_nv, x_ok, h0, h1, h2);
        *reinterpret_cast<float4*>(&DR_XS(buf)[0][c_pix][c_ch]) = h0;
        *reinterpret_cast<float4*>(&DR_XS(buf)[1][c_pix][c_ch]) = h1;
        *reinterpret_cast<float4*>(&DR_XS(buf)[2][c_pix][c_ch]) = h2;
        split8(ga, gb, g_nv, g_ok, h0, h1, h2);
        *reinterpret_cast<float4*>(&DR_GS(buf)[0][c_pix][c_ch]) = h0;
        *reinterpret_cast<float4*>(&DR_GS(buf)[1][c_pix][c_ch]) = h1;
        *reinterpret_cast<float4*>(&DR_GS(buf)[2][c_pix][c_ch]) = h2;
    };

    dr_f32x16 acc[TM][TM], lo[TM][TM];                                    // leading products / the five corrections (conv_x3.h)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = lo[i][j][r] = 0.f;

    if (steps > 0) {
        load();
        store(0);
    }
    __syncthreads();
    const int na_ = (p.Cin - (ci0 + wm * WT) + 31) / 32, nb_ = (p.Cout - (co0 + wn * WT) + 31) / 32;
    const int na = na_ < 0 ? 0 : (na_ > TM ? TM : na_), nb = nb_ < 0 ? 0 : (nb_ > TM ? TM : nb_);
    // transpose-read addressing (conv_wgrad_tr.h): lane = 16 * grp + i16; grp & 1 selects the 16-channel half of the 32-channel MFMA
    // tile, grp >> 1 = lk the k half; the lane SUPPLIES chunk i16 of the [4 pixels][16 channels] block
    const int i16 = lane & 15, half = (lane >> 4) & 1;
    const int t_pix = 8 * lk + (i16 >> 2);
    const int t_ch = 16 * half + 4 * (i16 & 3);
    auto k_step = [&](const int buf, const bool more) __attribute__((always_inline)) {
        if (more) load();
        dr_bf16x8 a[3][TM], b[3][TM];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const uint2 a0 = dr_lds_read_tr16(&DR_XS(buf)[pl][t_pix][wm * WT + 32 * t + t_ch]);
                const uint2 a1 = dr_lds_read_tr16(&DR_XS(buf)[pl][t_pix + 4][wm * WT + 32 * t + t_ch]);
                const uint2 b0 = dr_lds_read_tr16(&DR_GS(buf)[pl][t_pix][wn * WT + 32 * t + t_ch]);
                const uint2 b1 = dr_lds_read_tr16(&DR_GS(buf)[pl][t_pix + 4][wn * WT + 32 * t + t_ch]);
                a[pl][t] = __builtin_bit_cast(dr_bf16x8, make_float4(__builtin_bit_cast(float, a0.x), __builtin_bit_cast(float, a0.y),
                                                                       __builtin_bit_cast(float, a1.x), __builtin_bit_cast(float, a1.y)));
                b[pl][t] = __builtin_bit_cast(dr_bf16x8, make_float4(__builtin_bit_cast(float, b0.x), __builtin_bit_cast(float, b0.y),
                                                                       __builtin_bit_cast(float, b1.x), __builtin_bit_cast(float, b1.y)));
            }
        auto mf = [&](auto NA, auto NB) __attribute__((always_inline)) {
#define X3W_MMA(c, pa, pb)                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < decltype(NA)::value; ++i) _Pragma("unroll") for (int j = 0; j < decltype(NB)::value; ++j) \
        c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa][i], b[pb][j], c[i][j], 0, 0, 0)
            X3W_MMA(lo, 2, 0); X3W_MMA(lo, 0, 2); X3W_MMA(lo, 1, 1); X3W_MMA(lo, 1, 0); X3W_MMA(lo, 0, 1);
            X3W_MMA(acc, 0, 0);
#undef X3W_MMA
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
    float* dst = p.partial + ((long)split * taps + tap) * (p.cin_total ? p.cin_total : p.Cin) * p.Cout;      // (cin_total: conv_wgrad.h, the tail split)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int co = co0 + wn * WT + 32 * j + li;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
                const int ci = ci0 + wm * WT + 32 * i + row;
                if (ci < p.Cin && co < p.Cout) DR_NT_STORE(16, &dst[(long)ci * p.Cout + co], acc[i][j][r] + lo[i][j][r]);
            }
    }
#undef DR_XS
#undef DR_GS
}

}  // namespace dr
