// conv_x3h.h -- conv_x3_kernel for 3x3 layers with the input tile RESIDENT in LDS across the nine taps.
//
// Same operator, arithmetic and bits as conv_x3_kernel (fp32-accurate products on the bf16 matrix cores, conv_x3.h; tf.nn.conv2d of
// network/slim/ops.py:282 with the fused epilogue of conv_epilogue.inc).  conv_x3_kernel treats a 3x3 convolution as nine 1x1 ones:
// every K-tile (16 channels of one tap) fetches its 128 pixels again, splits them into the three bf16 planes again and writes them to
// LDS again -- 9 x the loads, the ~40-instruction split and the ds_writes (profiles/r05_conv_x3_sq_counters.md: 6.7 VALU instructions
// per MFMA).  Here a workgroup's 128 output pixels are whole image rows (W divides 128), so the pixels its nine taps read are ONE
// haloed tile of (128 / W + 2) x (W + 2) pixels: per 16-channel chunk it is fetched, split and stored once (1.6 float4 units per
// thread at W = 32 instead of 9; one unit at a time through one staging register: fetched in tap 0 / 2 / 4 / 6, stored two taps later) and the nine taps read their A fragments from it at a shifted row -- the shift is address
// arithmetic on the fragment read (five VALU instructions per 32-row group and tap), TF's 'SAME' zero padding is zeros stored in the
// halo.  The weight tile of every tap (pre-split planes, conv_x3.h) goes L2 -> LDS by LDS-DMA (conv_p3.h: inline asm with counted waits,
// two stages): no staging registers -- with them hipcc, at the 128-register limit of four waves per SIMD, sank each weight load to
// just before its ds_write and waited for it there.
//
//   LDS: 2 x halo [pixels][3 planes x 32 B + 16 B pad] (double-buffered per chunk: chunk c+1 is fetched and stored during taps 1 .. 7 of
//   chunk c) + 2 x weights [3 planes][BN][32 B] = 70 KB at W = 32, BN = 128: two workgroups per CU.  A halo pixel is 112 bytes: the 16
//   lanes of a ds_read_b128 group read 16 CONSECUTIVE halo pixels whatever the tap's shift, and 16 consecutive multiples of 28 dwords
//   are 16 different multiples of 4 modulo 64 -- every bank once, no slot swizzle, so a tap's shift is a constant byte offset: with
//   the image width a template parameter it is the immediate of the ds_read, and a tap costs no address arithmetic at all.
#pragma once
#include "conv_p3.h"

namespace dr {

// LW = log2(image width): 5 (the 32x32 maps of every configuration at 128x128 input) or 4.  Tiles as conv_x3_kernel's: 128 columns = eight
// waves of 64x32 (NW = 8, WM_ = 2), 64 columns = four waves of 64x32 (NW = 4, WM_ = 2), 96 columns (the 65..96-channel layers) = four
// waves of 32x96 that split the rows only (NW = 4, WM_ = 4).
template <int BN, int LW, int NW = 8, int WM_ = 2>
__global__ __launch_bounds__(NW * 64, (NW == 8 ? 4 : 2)) void conv_x3h_kernel(const ConvParams p) {
    constexpr int BM = 128, NT = NW * 64, WM = WM_, WN = NW / WM_, MF = 32, ABL = 0, CK = 16;
    constexpr int kWTM = BM / WM, kWTN = BN / WN, kTM = kWTM / 32, kTN = kWTN / 32;
    static_assert(kWTM % 32 == 0 && kWTN % 32 == 0, "wave tile of whole 32x32 MFMA tiles");
    constexpr int W = 1 << LW, Wp = W + 2, R = BM / W, NH = (R + 2) * Wp;      // tile = R whole image rows; NH halo pixels
    static_assert(W <= 32 && BM % W == 0, "a 32-lane fragment group reads whole runs of 16 consecutive pixels");
    constexpr int PIX = 112;                                   // bytes per halo pixel: [3 planes][16 bf16] + 16 B pad
    constexpr int AH = NH * PIX;                               // bytes per halo buffer
    constexpr int AI = (NH * 4 + NT - 1) / NT;                 // float4 staging units per thread and chunk
    constexpr int BP = BN * 32, BS = 3 * BP;                   // bytes per weight plane / per weight stage
    constexpr int kBUnits = 3 * BN * 2;
    constexpr int kBInstr = kBUnits / 64;                      // 1 KB LDS-DMA copies per weight stage: three per wave, waves 0 ..
    static_assert(AH % 16 == 0 && AI <= 4 && kBUnits % 64 == 0 && kBInstr <= 3 * NW, "halo buffer / weight copies");
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * AH + 2 * BS];

    DR_PIN_ARGS(p.x, p.x_cs, p.x_coff, p.Cin, p.B, p.H, p.w3, p.Kp, p.Np, p.zeros, p.nfast, p.gx, p.gy, p.Ng);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: LDS-DMA bases live in M0)
    const int wm = wave / WN, wn = wave % WN;
    const int H = p.H, HW = H * W;
    const int M = p.B * HW;
    const int gx = p.gx, gy = p.gy;
    int mblk = blockIdx.x, nblk = blockIdx.y;                  // the XCD-aware mapping of conv_igemm.h
    if (p.nfast && gy > 1) {
        const int L = blockIdx.y * gx + blockIdx.x, nN = gy;
        if ((gx & 7) == 0) { const int s = L >> 3; mblk = (L & 7) * (gx >> 3) + s / nN; nblk = s % nN; }
        else { mblk = L / nN; nblk = L % nN; }
    } else if ((gx & 7) == 0) {
        mblk = (blockIdx.x & 7) * (gx >> 3) + (blockIdx.x >> 3);
    }
    const int m0 = mblk * BM, n0 = nblk * BN;
    const int KT = (p.Kp + CK - 1) / CK;
    const int img = m0 / HW, y0 = (m0 - img * HW) >> LW;       // image and first row of the tile (launcher: H * W is a multiple of 128)

    // ---- halo staging: unit (h, q) = channels 4q .. 4q+3 of halo pixel h; same units for every chunk -----------------------------
    // (kDead in a_off: the pixel lies outside the image -- zeros are stored; in a_lds: the unit lies outside the halo -- nothing is stored)
    constexpr unsigned kDead = 0xFFFFFFFFu;
    unsigned a_off[AI], a_lds[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int u = tid + i * NT, h = u >> 2, q = u & 3;
        const bool live = h < NH;
        const int hh = live ? h : 0;
        const int hy = hh / Wp, hx = hh - hy * Wp;
        const int y = y0 + hy - 1, x = hx - 1;
        const bool ok = live && y >= 0 && y < H && x >= 0 && x < W;
        a_off[i] = ok ? (unsigned)(((long)(img * H + y) * W + x) * p.x_cs + p.x_coff + q * 4) : kDead;
        a_lds[i] = live ? (unsigned)(hh * PIX + q * 8) : kDead;
    }
    const int a_q4 = (tid & 3) * 4;                            // first channel of this thread's units within a chunk (NT % 4 == 0)
    static_assert(NT % 4 == 0, "a thread's units share their channel group");
    // weight planes (conv_x3.h: [Kp/16][tap][Np][3][16] in HBM) by LDS-DMA: a stage is [plane][BN rows][2 slots of 16 B] = kBInstr copies
    // of 1 KB; wave w issues copies 3 w .. 3 w + 2 (those that exist).  Lane L of copy q owns unit u = 64 q + L = (plane, row, physical
    // slot) and fetches the logical slot physical ^ ((row >> 3) & 1) (the swizzle of conv_x3.h on the source side); rows beyond Np offer an
    // out-of-range offset: zeros.
    const int b_n = kBInstr - wave * 3 < 0 ? 0 : (kBInstr - wave * 3 > 3 ? 3 : kBInstr - wave * 3);     // copies of this wave (uniform)
    const P3Src srcB = p3_src(p.w3, 0, (size_t)9 * KT * p.Np * 96);
    unsigned b_voff[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int u = (wave * 3 + j) * 64 + lane;
        const int pl = u / (BN * 2), within = u % (BN * 2), row = within >> 1, ls = (within & 1) ^ ((row >> 3) & 1);
        b_voff[j] = (u < kBUnits && n0 + row < p.Np) ? (unsigned)(((n0 + row) * 3 + pl) * 32 + ls * 16) : kP3Oob;
    }
    const unsigned b_dst = (unsigned)(2 * AH + wave * 3 * 1024);
    unsigned b_soff = 0;
    const unsigned w_tile = (unsigned)p.Np * 96u;

    float4 a_reg;                                               // one staging register set: unit i is fetched in tap 2 i and stored in tap 2 i + 2
    const bool ragged = (p.Cin & 3) != 0;
    auto load_a = [&](const int i, const int kc) __attribute__((always_inline)) {
        const bool ok = a_off[i] != kDead && kc + a_q4 < p.Cin;
        a_reg = *reinterpret_cast<const float4*>(ok ? p.x + kc + a_off[i] : p.zeros);
    };
    auto store_a = [&](const int i, unsigned char* dst, const int kc) __attribute__((always_inline)) {
        if (a_lds[i] == kDead) return;
        float4 v = a_reg;
        if (ragged && kc + CK > p.Cin) {
            const int left = p.Cin - (kc + a_q4);                           // (<= 0: the unit was loaded from the zero page)
            v.y = left > 1 ? v.y : 0.f;
            v.z = left > 2 ? v.z : 0.f;
            v.w = left > 3 ? v.w : 0.f;
        }
        uint2 h0, h1, h2;
        x3_split4(v, h0, h1, h2);
        *reinterpret_cast<uint2*>(dst + a_lds[i]) = h0;
        *reinterpret_cast<uint2*>(dst + a_lds[i] + 32) = h1;
        *reinterpret_cast<uint2*>(dst + a_lds[i] + 64) = h2;
    };
    auto dma_b = [&](const unsigned stage) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < b_n) p3_dma16(srcB, b_voff[j], b_soff, lds, b_dst + stage * BS + j * 1024);
        b_soff += w_tile;
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
    // this lane's A rows in the halo: row r = wm * 64 + i * 32 + li of the tile is image row r / W, column r % W, halo pixel
    // (r / W + 1) * Wp + r % W + 1 at the centre tap; the base is biased by the most negative tap shift (Wp + 1 pixels) so that every
    // tap's shift is a non-negative immediate
    unsigned a_frag[kTM];
#pragma unroll
    for (int i = 0; i < kTM; ++i) {
        const int r = wm * kWTM + i * 32 + li;
        a_frag[i] = (unsigned)((((r >> LW) + 1) * Wp + (r & (W - 1)) + 1 - (Wp + 1)) * PIX + lk * 16);
    }
    const unsigned b_frag = (unsigned)((wn * kWTN + li) * 32 + (lk ^ ((li >> 3) & 1)) * 16);

    unsigned char* const a_buf0 = lds;
    const unsigned char* const b_buf0 = lds + 2 * AH;
#pragma unroll
    for (int i = 0; i < AI; ++i) { load_a(i, 0); store_a(i, a_buf0, 0); }
    dma_b(0u);
    P3_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
#define X3H_MMA(c, a, b)                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < kTM; ++i) _Pragma("unroll") for (int j = 0; j < kTN; ++j)                              \
        c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dr_bf16x8, a[i]), __builtin_bit_cast(dr_bf16x8, b[j]), c[i][j], 0, 0, 0)
    unsigned bpar = 0;                                                      // weight stage of the tile being multiplied
    for (int c = 0; c < KT; ++c) {
        const unsigned char* const ah = lds + (c & 1) * AH;
        unsigned char* const ah_next = lds + ((c & 1) ^ 1) * AH;
        const bool more_c = c + 1 < KT;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const bool more = more_c || tap < 8;
            if (more) dma_b(bpar ^ 1u);                                   // (that stage was last read in the previous tap: behind its barrier)
            {   // one K-tile: the products in conv_x3_kernel's order (the same bits)
                constexpr int kCentre = Wp + 1;
                const int shift = ((tap / 3 - 1) * Wp + (tap % 3 - 1) + kCentre) * PIX;      // (a constant of the unrolled tap)
                const unsigned char* ap[kTM];
#pragma unroll
                for (int i = 0; i < kTM; ++i) ap[i] = ah + a_frag[i] + shift;
                const unsigned char* const bp = b_buf0 + bpar * BS + b_frag;
                float4 a0[kTM], b0[kTN], ax[kTM], bx[kTN];
#define X3H_READ_A(d, pl) _Pragma("unroll") for (int i = 0; i < kTM; ++i) d[i] = *reinterpret_cast<const float4*>(ap[i] + (pl) * 32)
#define X3H_READ_B(d, pl) _Pragma("unroll") for (int j = 0; j < kTN; ++j) d[j] = *reinterpret_cast<const float4*>(bp + (pl) * BP + j * 32 * 32)
                X3H_READ_A(a0, 0); X3H_READ_B(b0, 0); X3H_READ_A(ax, 2); X3H_READ_B(bx, 2);
                X3H_MMA(lo, ax, b0);                                        // a2*b0
                X3H_MMA(lo, a0, bx);                                        // a0*b2
                X3H_READ_A(ax, 1); X3H_READ_B(bx, 1);
                X3H_MMA(acc, a0, b0);
                X3H_MMA(lo, ax, b0);                                        // a1*b0
                X3H_MMA(lo, a0, bx);                                        // a0*b1
                X3H_MMA(lo, ax, bx);                                        // a1*b1
#undef X3H_READ_A
#undef X3H_READ_B
            }
            // the halo of chunk c + 1, one unit at a time through a_reg: stored two taps after it was requested, then the next one requested
            if (!(tap & 1) && tap >= 2 && (tap - 2) / 2 < AI && more_c) store_a((tap - 2) / 2, ah_next, (c + 1) * CK);
            const bool fetch_a = !(tap & 1) && tap / 2 < AI && more_c;
            if (fetch_a) load_a(tap / 2, (c + 1) * CK);
            bpar ^= 1;
            // the weight copy issued at the top of this tap has landed (this wave's share; the halo fetch just issued may stay in flight),
            // every LDS read and write of this wave has completed -- then everybody's
            __builtin_amdgcn_sched_barrier(0);                              // (the tap's MFMAs are issued BEFORE the wait: the copy lands under them)
            if (fetch_a) P3_WAIT_VM(1); else P3_WAIT_VM(0);
            __builtin_amdgcn_s_barrier();
        }
    }
#undef X3H_MMA
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
    constexpr int EP_BATCH_ROWS = NW == 8 ? 4 : 8;
    constexpr int EP_TS = MF, EP_NR = NR;
    const int ep_lg = lk, ep_lc = li;
    {
        constexpr bool EP_Y16 = false, EP_B16 = false, EP_B16_CONST = false;
#include "conv_epilogue.inc"
    }
    if (p.stat_part) {
        double* red = reinterpret_cast<double*>(lds);                      // the operand tiles are dead: the K loop ended on a barrier
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
    (void)NW; (void)M;
}

}  // namespace dr
