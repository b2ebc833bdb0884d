// conv_p3.h -- the implicit-GEMM convolution of conv_x3.h (fp32-accurate products on the bf16 matrix cores) on an input that is
// STORED as its three bf16 planes: no operand split, no staging registers and no ds_write in the K loop.
//
// Same operator, same arithmetic and the same bits as conv_x3_kernel (tf.nn.conv2d + BatchReNorm/bias + ReLU + residual + dropout:
// network/slim/ops.py:219-299, network/um_v1.py:18-48): six plane products per 16 k on v_mfma_f32_32x32x16_bf16, the leading one
// into `acc`, the five corrections into `lo`, the fused epilogue of conv_epilogue.inc.  What changes is where the split happens:
// the PRODUCER of the tensor (the BatchReNorm apply passes of train_kernels.h, p3_split_kernel below) writes
//        v = v0 + v1 + v2,   v0 = bf16(v), v1 = bf16(v - v0), v2 = bf16(v - v0 - v1)          (round to nearest even, conv_x3.h)
// ONCE, as "P3" storage [M][Cp/16][3][16] bf16 (Cp = channels rounded up to 16, pad channels zero): the three planes of one
// 16-channel chunk of one pixel are 96 contiguous bytes -- exactly the A operand of one K-tile and one row.  conv_x3_kernel split every
// pixel once per (row block, column block, tap): 9 x Cout/128 times in a 3x3 layer, ~40 VALU instructions per four channels, 6.7
// VALU per MFMA in the loop (profiles/r05_conv_x3_sq_counters.md).  Here both operand tiles go HBM/L2 -> LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds), through a THREE-stage ring with counted waits: tile t+2 is in flight while tile t is multiplied.
//
//   LDS stage = A [128 rows][3 planes][32 B] | B [BN rows][3 planes][32 B]  (96-byte rows; the 16-byte slot s of (row, plane) lives at
//   s ^ ((row >> 3) & 1): the 16 lanes of a ds_read_b128 group cover 8 row phases x 2 slots = all 64 banks once)
//   A DMA instruction moves 1 KB = 64 lanes x 16 B to LDS bytes [base, base + 1024): the LDS image is lane-linear, so lane L of
//   instruction q owns unit u = 64 q + L -> (row u / 6, plane (u % 6) / 2, physical slot u % 2) and fetches the LOGICAL slot
//   from its source.  Waves 0-3 copy the A tile (rows 32 w .. 32 w + 31: three instructions each), waves 4-7 the B tile.
//   Source addressing: one buffer descriptor per operand, a 32-bit per-lane byte offset fixed for the whole launch, and a scalar
//   offset that carries the K-tile (tap shift + channel chunk / weight tile).  A lane whose tap falls outside the image, beyond M,
//   or on a masked row offers an out-of-range offset: the hardware writes zeros to LDS (TF 'SAME' padding with no select on data).
//   Weights: the same three planes as conv_x3.h, [Kp/16][tap][Np][3][16] (pack_all_kernel): a B tile is one contiguous block.
//
// Synchronisation (every copy is inline asm, invisible to hipcc's wait-count insertion, so the counts are explicit):
//   iteration t:  issue DMA(t + 2) -> stage (t + 2) % 3   (last read in iteration t - 1: behind that iteration's barrier)
//                 fragments of tile t (9 ds_read_b128) and its 12 MFMAs
//                 s_waitcnt vmcnt(3): this wave's copies of tile t + 1 have landed (3 per wave and tile; t + 2's stay in flight)
//                 s_barrier: everybody's have -> tile t + 1 may be read
#pragma once
#include "conv_x3.h"

namespace dr {


// ---- P3 storage -------------------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int p3_cp(int C) { return (C + 15) & ~15; }
// bf16 elements of a P3 tensor of M rows and C channels
__host__ __device__ inline size_t p3_elems(long M, int C) { return (size_t)M * (size_t)p3_cp(C) * 3; }
// bf16-element index of (row m, channel c, plane pl)
__host__ __device__ inline long p3_index(long m, int c, int pl, int Cp) { return m * (long)Cp * 3 + (long)(c >> 4) * 48 + pl * 16 + (c & 15); }

// fp32 view [M][x_cs] (channels x_coff .. x_coff + C) -> P3 [M][Cp/16][3][16]; one thread = four channels of a row, pads written as zeros
__global__ __launch_bounds__(256) void p3_split_kernel(const float* x, int x_cs, int x_coff, int C, long M, __bf16* out, int Cp) {
    const int c4n = Cp / 4;
    const long total = M * c4n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / c4n;
        const int c = (int)(i % c4n) * 4;
        const float* src = x + m * x_cs + x_coff + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c + 4 <= C && ((x_cs | x_coff) & 3) == 0) v = *reinterpret_cast<const float4*>(src);
        else {
            if (c + 0 < C) v.x = src[0];
            if (c + 1 < C) v.y = src[1];
            if (c + 2 < C) v.z = src[2];
            if (c + 3 < C) v.w = src[3];
        }
        uint2 h0, h1, h2;
        x3_split4(v, h0, h1, h2);
        __bf16* d = out + p3_index(m, c, 0, Cp);
        *reinterpret_cast<uint2*>(d) = h0;
        *reinterpret_cast<uint2*>(d + 16) = h1;
        *reinterpret_cast<uint2*>(d + 32) = h2;
    }
}

// (the LDS-DMA primitive p3_dma16 / P3Src / P3_WAIT_VM: x3_dma.h)

// The kernel.  BM = 128 rows, BN = 128 columns, eight waves of 64x32 (conv_x3_kernel's default shape: 128 VGPRs, four waves per SIMD,
// two workgroups per CU with 72 KB of LDS each).
// VAR (experiments; the product is 0): bit 0 = the three copies of tile t+2 issued BETWEEN the MFMA groups of tile t instead of ahead of
// its fragment reads; bit 1 = no copies in the loop (ablation, wrong results); bit 2 = no waits / barriers in the loop (ablation)
template <int BN, int VAR = 0>
__global__ __launch_bounds__(512, 4) void conv_p3_kernel(const ConvParams p) {
    constexpr int BM = 128, NT = 512, NW = 8, WM = 2, WN = 4, MF = 32, ABL = 0;
    constexpr int kWTM = BM / WM, kWTN = BN / WN, kTM = kWTM / 32, kTN = kWTN / 32;
    static_assert(BN == 128, "eight waves of 64x32");
    constexpr int A_BYTES = BM * 96, B_BYTES = BN * 96, ST = A_BYTES + B_BYTES, NST = 3;
    constexpr int NA = A_BYTES / 1024, NB = B_BYTES / 1024;              // DMA instructions per tile and operand
    static_assert(NA == 12 && NB == 12, "three copies per wave and K-tile");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NST * ST];

    DR_PIN_ARGS(p.xp3, p.xp3_cp, p.Cin, p.B, p.H, p.W, p.ksize, p.w3, p.Kp, p.Np, p.rowmask, p.nfast, p.gx, p.gy, p.Ng);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int HW = p.H * p.W;
    const int M = p.B * HW;
    const int gx = p.gx, gy = p.gy;
    int mblk = blockIdx.x, nblk = blockIdx.y;                             // the XCD-aware mapping of conv_igemm.h
    if (p.nfast && gy > 1) {
        const int L = blockIdx.y * gx + blockIdx.x, nN = gy;
        if ((gx & 7) == 0) { const int s = L >> 3; mblk = (L & 7) * (gx >> 3) + s / nN; nblk = s % nN; }
        else { mblk = L / nN; nblk = L % nN; }
    } else if ((gx & 7) == 0) {
        mblk = (blockIdx.x & 7) * (gx >> 3) + (blockIdx.x >> 3);
    }
    const int m0 = mblk * BM, n0 = nblk * BN;
    const int taps = p.ksize * p.ksize;
    const int KT = p.Kp / 16;                                             // (Kp % 16 == 0: launcher)
    const int T_total = taps * KT;
    const int pad = p.ksize / 2;
    const unsigned rowB = (unsigned)p.xp3_cp * 6u;                        // bytes per P3 row
    const int bias_pix = pad ? p.W + 1 : 0;

    // ---- what this wave copies: three instructions of one operand, per-lane offsets fixed for the launch ---------------------------
    const bool a_wave = wave < 4;
    const P3Src srcA = p3_src(p.xp3, (long)bias_pix * rowB, (size_t)M * rowB);
    const P3Src srcB = p3_src(p.w3, 0, (size_t)T_total * p.Np * 96);
    P3Src src;
#if defined(DR_EMU)
    src = a_wave ? srcA : srcB;
#else
#pragma unroll
    for (int k = 0; k < 4; ++k) src.rsrc[k] = a_wave ? srcA.rsrc[k] : srcB.rsrc[k];
#endif
    unsigned voff[3], vtaps[3], dst_off[3];
    const bool pow2 = (p.W & (p.W - 1)) == 0 && (HW & (HW - 1)) == 0;
    const int w_shift = __builtin_ctz((unsigned)p.W);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int q = (wave & 3) * 3 + j;                                 // instruction of the operand tile: LDS bytes [1024 q, 1024 q + 1024)
        const int u = q * 64 + lane, row = u / 6, within = u % 6;
        const int pl = within >> 1, ls = (within & 1) ^ ((row >> 3) & 1);
        dst_off[j] = (unsigned)((a_wave ? 0 : A_BYTES) + q * 1024);
        if (a_wave) {
            const int m = m0 + row;
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
            vtaps[j] = ok ? mask : 0u;
            voff[j] = (unsigned)mm * rowB + (unsigned)(pl * 32 + ls * 16);
        } else {
            const int n = n0 + row;
            vtaps[j] = n < p.Np ? 0x1FFu : 0u;
            voff[j] = (unsigned)(n < p.Np ? n : 0) * 96u + (unsigned)(pl * 32 + ls * 16);
        }
    }
    // cursor of the NEXT tile to copy (taps innermost: conv_igemm.h)
    int ld_tap = 0, ld_dy = -pad, ld_dx = -pad, ld_kc = 0;
    unsigned soffB = 0;
    const unsigned wtile = (unsigned)p.Np * 96u;
    unsigned cur_soff = 0;
    auto issue_one = [&](const unsigned st_off, const int j) __attribute__((always_inline)) {
        if (j == 0) {
            const unsigned soffA = (unsigned)(ld_dy * p.W + ld_dx + bias_pix) * rowB + (unsigned)ld_kc * 6u;
            cur_soff = a_wave ? soffA : soffB;
        }
        const unsigned v = ((vtaps[j] >> ld_tap) & 1u) ? voff[j] : kP3Oob;
        p3_dma16(src, v, cur_soff, lds, st_off + dst_off[j]);
    };
    auto issue_adv = [&]() __attribute__((always_inline)) {
        soffB += wtile;
        ++ld_tap;
        if (++ld_dx > pad) { ld_dx = -pad; ++ld_dy; }
        if (ld_tap == taps) { ld_tap = 0; ld_dy = ld_dx = -pad; ld_kc += 16; }
    };
    auto issue = [&](const unsigned st_off) __attribute__((always_inline)) {
        const unsigned soffA = (unsigned)(ld_dy * p.W + ld_dx + bias_pix) * rowB + (unsigned)ld_kc * 6u;
        const unsigned soff = a_wave ? soffA : soffB;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const unsigned v = ((vtaps[j] >> ld_tap) & 1u) ? voff[j] : kP3Oob;
            p3_dma16(src, v, soff, lds, st_off + dst_off[j]);
        }
        soffB += wtile;
        ++ld_tap;
        if (++ld_dx > pad) { ld_dx = -pad; ++ld_dy; }
        if (ld_tap == taps) { ld_tap = 0; ld_dy = ld_dx = -pad; ld_kc += 16; }
    };

    using AccT = dr_f32x16;
    constexpr int NR = 16;
    AccT acc[kTM][kTN], lo[kTM][kTN];
#pragma unroll
    for (int i = 0; i < kTM; ++i)
#pragma unroll
        for (int j = 0; j < kTN; ++j)
#pragma unroll
            for (int r = 0; r < NR; ++r) { acc[i][j][r] = 0.f; lo[i][j][r] = 0.f; }

    const int lk = lane >> 5, li = lane & 31;
    const int fslot = lk ^ ((li >> 3) & 1);
    const unsigned a_frag = (unsigned)((wm * kWTM + li) * 96 + fslot * 16);
    const unsigned b_frag = (unsigned)(A_BYTES + (wn * kWTN + li) * 96 + fslot * 16);
#define P3_READ_A(d, pl) _Pragma("unroll") for (int i = 0; i < kTM; ++i) d[i] = *reinterpret_cast<const float4*>(a_ptr + i * 32 * 96 + (pl) * 32)
#define P3_READ_B(d, pl) _Pragma("unroll") for (int j = 0; j < kTN; ++j) d[j] = *reinterpret_cast<const float4*>(b_ptr + j * 32 * 96 + (pl) * 32)
#define P3_MMA(c, a, b)                                                                                                          \
    _Pragma("unroll") for (int i = 0; i < kTM; ++i) _Pragma("unroll") for (int j = 0; j < kTN; ++j)                              \
        c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dr_bf16x8, a[i]), __builtin_bit_cast(dr_bf16x8, b[j]), c[i][j], 0, 0, 0)
    // The stage is a RUN-TIME offset (two address adds per K-tile, every fragment read "base + immediate"): with compile-time
    // stages the loop is unrolled by three and its 0..2 leftover tiles are copies of the body behind branches, where hipcc
    // parked the accumulators in scratch (284 bytes per lane).
    // prologue: tiles 0 and 1 (T_total >= 2: launcher)
    issue(0u);
    issue((unsigned)ST);
    P3_WAIT_VM(3);
    __builtin_amdgcn_s_barrier();
    unsigned st_off = 0u, is_off = 2u * ST;                                 // stage of tile t / of tile t + 2
    for (int t = 0; t < T_total; ++t) {
        const bool more2 = (VAR & 2) ? false : t + 2 < T_total;
        if (!(VAR & 1) && more2) issue(is_off);
        {   // one K-tile (the products in conv_x3_kernel's order: the same bits)
            const unsigned char* a_ptr = lds + st_off + a_frag;
            const unsigned char* b_ptr = lds + st_off + b_frag;
            float4 a0[kTM], b0[kTN], ax[kTM], bx[kTN];
            P3_READ_A(a0, 0); P3_READ_B(b0, 0); P3_READ_A(ax, 2); P3_READ_B(bx, 2);
            P3_MMA(lo, ax, b0);                                             // a2*b0
            if ((VAR & 1) && more2) { __builtin_amdgcn_sched_barrier(0); issue_one(is_off, 0); __builtin_amdgcn_sched_barrier(0); }
            P3_MMA(lo, a0, bx);                                             // a0*b2
            P3_READ_A(ax, 1); P3_READ_B(bx, 1);
            if ((VAR & 1) && more2) { __builtin_amdgcn_sched_barrier(0); issue_one(is_off, 1); __builtin_amdgcn_sched_barrier(0); }
            P3_MMA(acc, a0, b0);
            if ((VAR & 1) && more2) { __builtin_amdgcn_sched_barrier(0); issue_one(is_off, 2); issue_adv(); __builtin_amdgcn_sched_barrier(0); }
            P3_MMA(lo, ax, b0);                                             // a1*b0
            P3_MMA(lo, a0, bx);                                             // a0*b1
            P3_MMA(lo, ax, bx);                                             // a1*b1
        }
        if (!(VAR & 4)) {
            if (more2) P3_WAIT_VM(3); else P3_WAIT_VM(0);                   // tile t + 1 has landed (this wave's share of it)
            __builtin_amdgcn_s_barrier();                                   // ... everybody's; and everybody is done reading tile t
        }
        st_off = st_off == 2u * ST ? 0u : st_off + ST;
        is_off = is_off == 2u * ST ? 0u : is_off + ST;
    }
#undef P3_READ_A
#undef P3_READ_B
#undef P3_MMA
#pragma unroll
    for (int i = 0; i < kTM; ++i)
#pragma unroll
        for (int j = 0; j < kTN; ++j) acc[i][j] += lo[i][j];

    // ---- epilogue: conv_epilogue.inc (the fp32 copy), as conv_x3_kernel ------------------------------------------------------------
    double s1[kTN], s2[kTN];
#pragma unroll
    for (int j = 0; j < kTN; ++j) s1[j] = s2[j] = 0.0;
    constexpr int EP_TM = kTM, EP_TN = kTN;
    const int ep_m0 = m0 + wm * kWTM, ep_n0 = n0 + wn * kWTN;
    const unsigned ep_rows = 0xFFFFu;
    constexpr int EP_BATCH_ROWS = 4;
    constexpr int EP_TS = MF, EP_NR = NR;
    const int ep_lg = lk, ep_lc = li;
    {
        constexpr bool EP_Y16 = false, EP_B16 = false, EP_B16_CONST = false;
#include "conv_epilogue.inc"
    }
    if (p.stat_part) {
        double* red = reinterpret_cast<double*>(lds);
        static_assert(sizeof(lds) >= sizeof(double) * 2 * WM * BN, "stat scratch");
#pragma unroll
        for (int j = 0; j < kTN; ++j) {
            double a = s1[j], b = s2[j];
            a += __shfl_xor(a, 32);
            b += __shfl_xor(b, 32);
            if (ep_lg == 0) {
                const int col = wn * kWTN + j * MF + ep_lc;
                red[(0 * WM + wm) * BN + col] = a;
                red[(1 * WM + wm) * BN + col] = b;
            }
        }
        __syncthreads();
        for (int e = tid; e < 2 * BN; e += NT) {
            const int which = e / BN, col = e % BN, n = n0 + col;
            double tsum = 0.0;
#pragma unroll
            for (int w = 0; w < WM; ++w) tsum += red[(which * WM + w) * BN + col];
            if (n < p.Cout) p.stat_part[((long)which * p.Cout + n) * gx + mblk] = tsum;
        }
    }
    (void)NW;
}

}  // namespace dr
