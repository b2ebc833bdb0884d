// conv_splitk.h -- the implicit-GEMM convolution for grids that cannot fill the chip (every layer below 32x32 pixels).
//
// On such a layer a 64x64 tile leaves most CUs idle (8x8 pixels, B = 40: 40 workgroups on 256 CUs) and each of its four
// waves walks the WHOLE K axis for one 32x32 MFMA tile: 3x3 64->64 is 288 dependent-rate v_mfma_f32_32x32x2 per wave,
// 18 k cycles = 9 us whatever the image size (measured 14-19 us per launch from 16x16 down to 2x2).  Here a workgroup
// owns ONE 32x32 output tile and its four waves split K: wave w takes K-tiles w, w+4, ... (a K-tile = 16 input
// channels of one tap, 32 with bf16 operands), stages them through its own LDS slice with its own two-deep pipeline
// -- no workgroup barrier inside the K loop, LDS operations of one wave execute in order -- and the four partial
// accumulators are summed through LDS at the end.  Four times as many workgroups, a quarter of the chain per wave.
// Each wave then runs the shared epilogue (conv_epilogue.inc) on a quarter of the tile's rows, so every fused
// feature of the tiled kernel (BatchReNorm statistics, residual, masks, dropout, backward sums) is available.
// Operand tiles, slot swizzle, weight packing and the bf16 variant are those of conv_igemm_kernel.
#pragma once
#include "conv_igemm.h"

namespace dr {

template <int BF, int ABL = 0, int IO = 0>      // IO: with the epilogue copies for bf16-stored raw outputs / gradients (conv_igemm.h, XB bit 1)
__global__ __launch_bounds__(256, 2) void conv_splitk_kernel(const ConvParams p) {
    DR_PIN_ARGS(p.x, p.x_cs, p.x_coff, p.Cin, p.B, p.H, p.W, p.ksize, p.w, p.Kp, p.Np, p.rowmask, p.zeros, p.gx);
    constexpr int BM = 32, BN = 32;
    constexpr int CK = BF ? 32 : 16;      // input channels per K-tile
    constexpr int CS = BF ? 8 : 4;        // input channels per 16-byte LDS slot
    constexpr int BKC = 16;               // floats per packed weight row
    __shared__ __attribute__((aligned(16))) float As[4][2][BM][16];
    __shared__ __attribute__((aligned(16))) float Bs[4][2][BN][16];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lk = lane >> 5, li = lane & 31;
    const int HW = p.H * p.W;
    const int M = p.B * HW;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int taps = p.ksize * p.ksize;
    const int KT = (p.Kp + CK - 1) / CK;
    const int T_total = taps * KT;
    const int pad = p.ksize / 2;
    auto swz = [](int row) { return row & 12; };                          // slot ^ ((row >> 2) & 3), in floats

    // ---- this lane's two A slots and two B slots of every K-tile --------------------------------------
    // (named scalars, not arrays: hipcc keeps small arrays captured by the loader lambdas in scratch)
    const bool pow2 = (p.W & (p.W - 1)) == 0 && (HW & (HW - 1)) == 0;
    const int w_shift = __builtin_ctz((unsigned)p.W);
    auto slot_setup = [&](const int i, unsigned& off, unsigned& tapmask) __attribute__((always_inline)) {
        const int s = lane + 64 * i;
        const int m = m0 + (s >> 2);
        bool ok = m < M;
        if (ok && p.rowmask) ok = !(p.rowmask[m] < p.mask_thresh);
        const int mm = ok ? m : 0;
        int y, x;
        if (pow2) { const int rem = mm & (HW - 1); y = rem >> w_shift; x = rem & (p.W - 1); }
        else { const int rem = mm % HW; y = rem / p.W; x = rem % p.W; }
        unsigned mask = 1u;
        if (p.ksize == 3) {
            const unsigned cols = (x > 0 ? 1u : 0u) | 2u | (x < p.W - 1 ? 4u : 0u);
            mask = (y > 0 ? cols : 0u) | (cols << 3) | (y < p.H - 1 ? cols << 6 : 0u);
        }
        tapmask = ok ? mask : 0u;
        off = ok ? (unsigned)((long)m * p.x_cs + p.x_coff + (s & 3) * CS) : 0u;
    };
    unsigned a_off0, a_off1, a_taps0, a_taps1;
    slot_setup(0, a_off0, a_taps0);
    slot_setup(1, a_off1, a_taps1);
    const bool ragged = (p.Cin & 3) != 0;
    // One K-tile's worth of staging registers.  THREE sets rotate, so three tiles of this wave are in flight: the chain is
    // "load -> LDS -> MFMA" per tile with nothing else to run on the SIMD (one wave per SIMD, grids of <= 160 workgroups), and
    // with a single set every tile paid a full memory round trip -- nine in a row for a 3x3 64->64 layer, ~10 us whatever the
    // image size.  (Separate named structs passed by reference to always-inline lambdas: arrays of them went to scratch.)
    struct TileRegs { float4 a0, a1, h0, h1, b0, b1; int nv0, nv1; };
    TileRegs R0, R1, R2;
    R0.h0 = R0.h1 = R1.h0 = R1.h1 = R2.h0 = R2.h1 = make_float4(0.f, 0.f, 0.f, 0.f);   // channels 4..7 of a slot: bf16 operands only
    R0.nv0 = R0.nv1 = R1.nv0 = R1.nv1 = R2.nv0 = R2.nv1 = CS;
    auto load_one = [&](const int i, const int kc, const int tap, const float* ld_x, const float* ld_w, const unsigned off,
                        const unsigned tapmask, float4& areg, float4& ahi, float4& breg, int& anv) __attribute__((always_inline)) {
        const int s = lane + 64 * i;
        const int left = p.Cin - (kc + (s & 3) * CS);
        const int nv = left < 0 ? 0 : (left > CS ? CS : left);
        const bool ok = ((tapmask >> tap) & 1u) && nv > 0;
        const float* src = ok ? ld_x + off : p.zeros;
        areg = *reinterpret_cast<const float4*>(src);
        if constexpr (BF) ahi = *reinterpret_cast<const float4*>(ok && nv > 4 ? src + 4 : p.zeros);
        anv = ok ? nv : CS;
        const int brow = s >> 2;
        breg = *reinterpret_cast<const float4*>(n0 + brow < p.Np ? ld_w + (unsigned)((n0 + brow) * BKC + (s & 3) * 4) : p.zeros);
    };
    auto load_tile = [&](const int t, TileRegs& R) __attribute__((always_inline)) {
        const int chunk = t / taps, tap = t - chunk * taps;
        const int kc = chunk * CK;
        const int ty = tap / p.ksize;
        const int dy = ty - pad, dx = tap - ty * p.ksize - pad;
        const float* ld_x = p.x + (long)(dy * p.W + dx) * p.x_cs + kc;
        const float* ld_w = p.w + (long)t * p.Np * BKC;                     // packed [chunk][tap][Np][16]: block t
        load_one(0, kc, tap, ld_x, ld_w, a_off0, a_taps0, R.a0, R.h0, R.b0, R.nv0);
        load_one(1, kc, tap, ld_x, ld_w, a_off1, a_taps1, R.a1, R.h1, R.b1, R.nv1);
    };
    auto store_one = [&](const int i, const int buf, float4 v, float4 u, const float4 breg, const int nv) __attribute__((always_inline)) {
        const int s = lane + 64 * i, r = s >> 2, k = (s & 3) * 4;
        if (ragged) {
            v.y = nv > 1 ? v.y : 0.f; v.z = nv > 2 ? v.z : 0.f; v.w = nv > 3 ? v.w : 0.f;
        }
        if constexpr (BF) {
            if (ragged) { u.y = nv > 5 ? u.y : 0.f; u.z = nv > 6 ? u.z : 0.f; u.w = nv > 7 ? u.w : 0.f; }
            const dr_f32x8 f = {v.x, v.y, v.z, v.w, u.x, u.y, u.z, u.w};
            v = __builtin_bit_cast(float4, __builtin_convertvector(f, dr_bf16x8));
        }
        *reinterpret_cast<float4*>(&As[wave][buf][r][k ^ swz(r)]) = v;
        *reinterpret_cast<float4*>(&Bs[wave][buf][r][k ^ swz(r)]) = breg;
    };
    auto store_tile = [&](const int buf, const TileRegs& R) __attribute__((always_inline)) {
        store_one(0, buf, R.a0, R.h0, R.b0, R.nv0);
        store_one(1, buf, R.a1, R.h1, R.b1, R.nv1);
    };

    dr_f32x16 acc0, acc1;                 // two independent MFMA chains, summed at the end
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;

    auto mfma_tile = [&](const int buf) __attribute__((always_inline)) {
        float4 a4[2], b4[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            a4[g] = *reinterpret_cast<const float4*>(&As[wave][buf][li][(g * 8 + lk * 4) ^ swz(li)]);
            b4[g] = *reinterpret_cast<const float4*>(&Bs[wave][buf][li][(g * 8 + lk * 4) ^ swz(li)]);
        }
        if constexpr (BF) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dr_bf16x8, a4[0]), __builtin_bit_cast(dr_bf16x8, b4[0]), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dr_bf16x8, a4[1]), __builtin_bit_cast(dr_bf16x8, b4[1]), acc1, 0, 0, 0);
        } else {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0].x, b4[0].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0].y, b4[0].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0].z, b4[0].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0].w, b4[0].w, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1].x, b4[1].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1].y, b4[1].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1].z, b4[1].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1].w, b4[1].w, acc1, 0, 0, 0);
        }
    };
    // Step k of this wave (tile t = wave + 4k, LDS buffer k & 1): the register set that tile k used is free again -> tile k+3 goes
    // into it; MFMAs over tile k; tile k+1 (loaded two steps ago) moves from its registers to the other LDS buffer.
    auto step = [&](const int t, const int buf, TileRegs& free_set, const TileRegs& next_set) __attribute__((always_inline)) {
        if (t + 12 < T_total) load_tile(t + 12, free_set);                 // wave-uniform
        mfma_tile(buf);
        if (t + 4 < T_total) store_tile(buf ^ 1, next_set);
        __builtin_amdgcn_wave_barrier();
    };
    {
        const int t0 = wave;
        if (t0 < T_total) load_tile(t0, R0);
        if (t0 + 4 < T_total) load_tile(t0 + 4, R1);
        if (t0 + 8 < T_total) load_tile(t0 + 8, R2);
        if (t0 < T_total) store_tile(0, R0);
        __builtin_amdgcn_wave_barrier();
        // six steps per round: register sets rotate with period 3, LDS buffers with period 2
        for (int t = t0; t < T_total; t += 24) {
            step(t, 0, R0, R1);
            if (t + 4 < T_total) step(t + 4, 1, R1, R2);
            if (t + 8 < T_total) step(t + 8, 0, R2, R0);
            if (t + 12 < T_total) step(t + 12, 1, R0, R1);
            if (t + 16 < T_total) step(t + 16, 0, R1, R2);
            if (t + 20 < T_total) step(t + 20, 1, R2, R0);
        }
    }

    // ---- sum the four K-partials: wave w keeps accumulator registers 4w .. 4w+3 (rows 8w + {0..3} + 4*lk) ------
    __syncthreads();                                                        // every wave is done with its operand tiles
    float* red = &As[0][0][0][0];                                           // [4 waves][16 regs][64 lanes] floats = 16 KB = sizeof(As)
    static_assert(sizeof(As) >= 4 * 16 * 64 * sizeof(float), "reduction scratch");
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc0[r] + acc1[r];
    __syncthreads();
    dr_f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float sacc = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) sacc += red[(w * 16 + wave * 4 + q) * 64 + lane];   // fixed order w = 0..3: deterministic
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
            if (wave == rr) acc[0][0][rr * 4 + q] = sacc;                               // wave-uniform select of the register
    }

    double s1[1], s2[1];
    s1[0] = s2[0] = 0.0;
    constexpr int EP_TM = 1, EP_TN = 1;
    const int ep_m0 = m0, ep_n0 = n0;
    const unsigned ep_rows = 0xFu << (4 * wave);
    constexpr int EP_BATCH_ROWS = 16;     // two waves per SIMD by launch bounds: room for the whole column in one batch
    constexpr int EP_TS = 32, EP_NR = 16;
    const int ep_lg = lk, ep_lc = li;
    if (IO != 0 && p.bst_raw_bf16) {                         // one copy of the epilogue per storage case (conv_epilogue.inc)
        constexpr bool EP_Y16 = false, EP_B16 = true, EP_B16_CONST = true;
#include "conv_epilogue.inc"
    } else if (IO != 0 && p.y_bf16) {
        constexpr bool EP_Y16 = true, EP_B16 = false, EP_B16_CONST = false;
#include "conv_epilogue.inc"
    } else {
        constexpr bool EP_Y16 = false, EP_B16 = false, EP_B16_CONST = false;
#include "conv_epilogue.inc"
    }

    if (p.stat_part) {
        __syncthreads();                                                    // the reduction scratch is dead
        double* sred = reinterpret_cast<double*>(&Bs[0][0][0][0]);          // [2][4 waves][32 columns]
        double a = s1[0], b = s2[0];
        a += __shfl_xor(a, 32);
        b += __shfl_xor(b, 32);
        if (lk == 0) {
            sred[(0 * 4 + wave) * BN + li] = a;
            sred[(1 * 4 + wave) * BN + li] = b;
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, col = tid % BN, n = n0 + col;
            double tsum = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) tsum += sred[(which * 4 + w) * BN + col];
            if (n < p.Cout) p.stat_part[((long)which * p.Cout + n) * p.gx + blockIdx.x] = tsum;
        }
    }
}

}  // namespace dr
