// conv_x3.h -- the implicit-GEMM convolution of conv_igemm.h with fp32-accurate products on the bf16 matrix cores.
//
// Same operator (tf.nn.conv2d + BatchReNorm/bias + ReLU + residual + dropout: network/slim/ops.py:219-299, network/um_v1.py:18-48),
// same tensors (fp32 NHWC in, fp32 out), same fused epilogue (conv_epilogue.inc).  What changes is how a product a*b is formed:
// every fp32 operand is split into three bf16 terms
//        a = a0 + a1 + a2,   a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)        (round to nearest even)
// which is EXACT to 2^-24 relative or better (three 8-bit significands cover fp32's 24 bits; each subtraction is exact), and
//        a*b  ~=  a0*b0 + a0*b1 + a1*b0 + a0*b2 + a1*b1 + a2*b0
// drops only the terms of relative size <= 2^-24 (a1*b2, a2*b1, a2*b2).  A bf16 x bf16 product is exact in fp32 and the matrix core
// accumulates in fp32, so the result carries the rounding of an fp32 accumulation plus ~2^-23 per product: the same error class as
// v_mfma_f32_32x32x2_f32 (tests: the unchanged 2e-5-of-range bar against the fp64 definition, and a direct comparison of both
// kernels' errors).  Six v_mfma_f32_32x32x16_bf16 (32 cycles each, 16 k) replace eight v_mfma_f32_32x32x2_f32 (64 cycles each,
// 2 k) per 16 k: 192 matrix-core cycles instead of 512 -- the arithmetic lever of the 3x3 and wide 1x1 layers.
//
//   GEMM view as conv_igemm.h:  M = B*H*W, N = Cout, K = taps*Cin, K-tile = 16 input channels of one tap
//   A: fp32 activations, split WHILE STAGED (one float4 per thread and 4 channels -> three 8-byte LDS writes)
//   B: weights arrive pre-split from pack_all_kernel: bf16 [Kp/16][tap][3 planes][Np][16]
//   LDS per stage and plane: [rows][16 bf16] = 32-byte rows of two 16-byte slots; lane (li = lane & 31, lk = lane >> 5) of a wave reads
//   the 8 k of slot lk of row li with one ds_read_b128 = its whole operand of one MFMA.  Slot s of row r lives at s ^ ((r >> 3) & 1):
//   every 16-lane group of a ds_read_b128 then covers all 64 banks once (MI355X_MICROARCH.md, LDS table).
//   Block = 256 threads = 2 x 2 waves; wave tile (BM/2) x (BN/2) of 32x32 MFMA tiles.  128x128: 12 fragment reads feed 24 MFMAs.
#pragma once
#include "conv_igemm.h"
#include "x3_dma.h"

namespace dr {

// four fp32 -> three planes of four bf16 (two words per plane), round to nearest even at every level.  hipcc turns the vector
// conversions into two v_cvt_pk_bf16_f32 per level for the planes PLUS four single conversions for the round trip back to fp32 (14
// conversions per call).  A hand-written form that reuses the packed words (shift / mask: 22 VALU instead of 38, -13 % VALU
// instructions in the SQ counters) measured EQUAL in the conv kernel and 8-11 % SLOWER in the weight-gradient kernel
// (profiles/r05_experiments.md section 8): kept behind DR_X3_SPLIT_HAND for the record, not built.
typedef float dr_f32x2 __attribute__((ext_vector_type(2)));
__host__ __device__ static inline float4 x3_as_f4(const float4& v) { return v; }
__host__ __device__ static inline float4 x3_as_f4(const dr_f32x4& v) { return make_float4(v[0], v[1], v[2], v[3]); }
typedef __bf16 dr_bf16x2 __attribute__((ext_vector_type(2)));
__host__ __device__ static inline unsigned x3_pack2(float a, float b) {
    const dr_f32x2 f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, dr_bf16x2));
}
__host__ __device__ static inline void x3_split4(const float4 v, uint2& h0, uint2& h1, uint2& h2) {
#if !defined(DR_X3_SPLIT_HAND)
    const dr_f32x4 f = {v.x, v.y, v.z, v.w};
    const dr_bf16x4 b0 = __builtin_convertvector(f, dr_bf16x4);
    const dr_f32x4 q1 = f - __builtin_convertvector(b0, dr_f32x4);
    const dr_bf16x4 b1 = __builtin_convertvector(q1, dr_bf16x4);
    const dr_f32x4 q2 = q1 - __builtin_convertvector(b1, dr_f32x4);
    const dr_bf16x4 b2 = __builtin_convertvector(q2, dr_bf16x4);
    h0 = __builtin_bit_cast(uint2, b0); h1 = __builtin_bit_cast(uint2, b1); h2 = __builtin_bit_cast(uint2, b2);
#else
    auto lo = [](unsigned w) { return __builtin_bit_cast(float, w << 16); };
    auto hi = [](unsigned w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); };
    h0 = make_uint2(x3_pack2(v.x, v.y), x3_pack2(v.z, v.w));
    const float r0 = v.x - lo(h0.x), r1 = v.y - hi(h0.x), r2 = v.z - lo(h0.y), r3 = v.w - hi(h0.y);
    h1 = make_uint2(x3_pack2(r0, r1), x3_pack2(r2, r3));
    h2 = make_uint2(x3_pack2(r0 - lo(h1.x), r1 - hi(h1.x)), x3_pack2(r2 - lo(h1.y), r3 - hi(h1.y)));
#endif
}

// RING = 1: three LDS stages instead of two.  With two, a K-tile ends "ds_write the next tile, barrier, ds_read the fragments, wait":
// the matrix cores idle through the barrier skew and an LDS round trip once per K-tile (768 cycles of MFMA work per wave).  With three,
// tile t+2 is written while tile t is multiplied, the barrier sits in the MIDDLE of the tile's MFMA sequence (waves wait for each other
// while their queued MFMAs execute), and the first fragments of tile t+1 -- in LDS since the previous barrier -- are fetched under the
// last MFMAs of tile t, so the next tile starts on registers that are already there.  The global loads run one tile further ahead.
// NW = 8: 512 threads = 2 x 4 waves, wave tile (BM/2) x (BN/4) -- half the accumulators per wave (two tiles instead of four), so twice
// the waves per SIMD fit the register file: more MFMA chains to interleave with the staging of the next tile.
// WM_ = 4 (with NW = 4): the waves split the rows only, each owns ALL BN columns -- the 96-column tile of the 65..96-channel layers
// (three column tiles per wave; a 2 x 2 layout would need 64-column multiples).
// BD = 1 (two-stage kernels): the weight planes of a K-tile go L2 -> LDS by an LDS-DMA hidden from hipcc (x3_dma.h: inline asm, explicit
// wait before a raw s_barrier) instead of through registers: no staging registers (the loads were sunk to their ds_write at the 128-register
// limit), no ds_write_b128, no 64-bit address selects -- the weight path of conv_x3h.h.
// BN = 256 (round 6, the wide 1x1 layers): the pixels of a row block are fetched and split once per 256 output columns instead of once
// per 128 -- as NW = 8 waves of 64x64 (four accumulator pairs: 256 registers, two waves per SIMD) or NW = 16 waves of 64x32 (the
// eight-wave kernel's wave, one 1024-thread workgroup per CU).
template <int BM, int BN, int LO, int NW> constexpr int x3_waves_per_simd() { return NW >= 8 ? (BM * BN / NW > 2048 ? 2 : 4) : LO ? 2 : 3; }
template <int BM, int BN, int LO = 1, int RING = 0, int NW = 4, int WM_ = 2, int BD = 0, int PF = 0, int ABL_ = 0>
__global__ __launch_bounds__(NW * 64, (x3_waves_per_simd<BM, BN, LO, NW>())) void conv_x3_kernel(const ConvParams p) {
    static_assert(BD == 0 || RING == 0, "the hidden weight copy belongs to the two-stage loop");
    static_assert(PF == 0 || BD == 1, "the two-tile pixel prefetch counts its waits by hand: the hidden weight copy only");
    // PF = 2: the two-stage loop as it is, plus a scheduling barrier between a K-tile's MFMAs and the split of the next tile's pixels: hipcc
    // otherwise hoists the split (and with it the wait for the global load issued a few instructions earlier) in among the first MFMAs of
    // every other K-tile -- the wave then sits out the whole HBM round trip with ten of its twelve MFMAs unissued.
    constexpr int NT = NW * 64;                              // threads
    constexpr int WM = WM_, WN = NW / WM_, WK = 1, MF = 32, ABL = (ABL_ & 3) == 3 ? 3 : 0;
    // ABL_ (measurement builds of the two-stage BD loop, debug library): 3 = no epilogue stores; after the first K-tile: 16 = no pixel
    // path (global loads, split, ds_write), 32 = no weight copies, 64 = no waits and no barrier; sums combine
    constexpr bool kNoA = (ABL_ & 16) != 0, kNoB = (ABL_ & 32) != 0, kNoSync = (ABL_ & 64) != 0;
    constexpr int kWTM = BM / WM, kWTN = BN / WN, kTM = kWTM / 32, kTN = kWTN / 32;
    static_assert(kWTM % 32 == 0 && kWTN % 32 == 0, "wave tile of whole 32x32 MFMA tiles");
    constexpr int CK = 16;                                   // input channels per K-tile
    constexpr int kAIters = (BM * 4 + NT - 1) / NT;          // float4 units (4 channels of a row) per thread
    constexpr int kBUnits = 3 * BN * 2;                      // 16-byte units of the three weight planes
    constexpr int kBIters = (kBUnits + NT - 1) / NT;
    // one LDS object per stage (conv_igemm.h: the wait-count insertion tells stages apart by object); [plane][row][2 slots] of 16 bytes
    __shared__ __attribute__((aligned(16))) float4 As0[3][BM][2];
    __shared__ __attribute__((aligned(16))) float4 As1[3][BM][2];
    __shared__ __attribute__((aligned(16))) float4 Bs0[3][BN][2];
    __shared__ __attribute__((aligned(16))) float4 Bs1[3][BN][2];
    __shared__ __attribute__((aligned(16))) float4 As2[3][BM][2];        // (RING only: an instantiation that never names them allocates nothing)
    __shared__ __attribute__((aligned(16))) float4 Bs2[3][BN][2];
#define DR_AS(stage) ((stage) == 0 ? As0 : (stage) == 1 ? As1 : As2)
#define DR_BS(stage) ((stage) == 0 ? Bs0 : (stage) == 1 ? Bs1 : Bs2)

    DR_PIN_ARGS(p.x, p.x_cs, p.x_coff, p.Cin, p.B, p.H, p.W, p.ksize, p.w3, p.Kp, p.Np, p.rowmask, p.zeros, p.nfast, p.gx, p.gy, p.Ng);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = BD ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;       // (BD: uniform -- LDS-DMA bases live in M0)
    const int wk = 0;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int HW = p.H * p.W;
    const int M = p.B * HW;
    // workgroup -> tile: the XCD-aware mapping of conv_igemm.h (row blocks of one XCD contiguous, N blocks of a row block back to back)
    const int gx = p.gx, gy = p.gy;
    int mblk = blockIdx.x, nblk = blockIdx.y;
    if (p.nfast && gy > 1) {
        const int L = blockIdx.y * gx + blockIdx.x, nN = gy;
        if ((gx & 7) == 0) { const int s = L >> 3; mblk = (L & 7) * (gx >> 3) + s / nN; nblk = s % nN; }
        else { mblk = L / nN; nblk = L % nN; }
    } else if ((gx & 7) == 0) {
        mblk = (blockIdx.x & 7) * (gx >> 3) + (blockIdx.x >> 3);
    }
    // Co-resident workgroups of equal K run in lock step: all of a round's workgroups reach their epilogues (HBM writes, no MFMA) and the
    // next round its prologues (HBM latency, no MFMA) together.  Delaying the second workgroup of every CU once, in the first round, by
    // about half a workgroup's life puts one of a CU's two in its K loop while the other stores / fetches (dispatch is breadth-first: of the
    // grid's first 512 workgroups, those numbered 32..63 within their XCD are the second on their CU).
#if !defined(DR_EMU)
    if (p.stagger > 0) {
        const unsigned L = blockIdx.y * gridDim.x + blockIdx.x;
        const int n = p.stagger & 0xFFFF, mode = p.stagger >> 16;
        bool late;
        if (mode == 0) late = ((L >> 3) >> 5) & 1u;                        // breadth-first dispatch: the XCD's workgroups 32..63
        else if (mode == 1) late = (L >> 3) & 1u;                          // depth-first: every other workgroup of an XCD
        else late = (__builtin_amdgcn_s_getreg(6148) & 15u) >= 2u;         // HW_ID.wave_id: this wave's slot on its SIMD (the first workgroup of a CU holds 0, 1)
        if (L < 512u && late)
            for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    const int m0 = mblk * BM;
    const int n0 = nblk * BN;
    const int taps = p.ksize * p.ksize;
    const int KT = (p.Kp + CK - 1) / CK;
    const int T_total = taps * KT;
    const int pad = p.ksize / 2;

    // ---- per-thread loader bookkeeping: unit (row, q) = channels 4q..4q+3 of the K-tile for output pixel m0 + row ------------------
    int a_row[kAIters], a_q[kAIters];
    unsigned a_off[kAIters], a_taps[kAIters];
    const bool pow2 = (p.W & (p.W - 1)) == 0 && (HW & (HW - 1)) == 0;
    const int w_shift = __builtin_ctz((unsigned)p.W);
#pragma unroll
    for (int i = 0; i < kAIters; ++i) {
        const int idx = tid + i * NT;
        const int row = idx >> 2;
        a_row[i] = row;
        a_q[i] = idx & 3;
        const int m = m0 + row;
        bool ok = row < BM && m < M;
        if (ok && p.rowmask) ok = !(p.rowmask[m] < p.mask_thresh);
        const int mm = ok ? m : 0;
        int y, x;
        if (pow2) {
            const int rem = mm & (HW - 1);
            y = rem >> w_shift; x = rem & (p.W - 1);
        } else {
            const int rem = mm % HW;
            y = rem / p.W; x = rem % p.W;
        }
        unsigned mask = 1u;
        if (p.ksize == 3) {
            const unsigned cols = (x > 0 ? 1u : 0u) | 2u | (x < p.W - 1 ? 4u : 0u);
            mask = (y > 0 ? cols : 0u) | (cols << 3) | (y < p.H - 1 ? cols << 6 : 0u);
        }
        a_taps[i] = ok ? mask : 0u;
        a_off[i] = ok ? (unsigned)((long)m * p.x_cs + p.x_coff + a_q[i] * 4) : 0u;
    }
    // weight planes: unit u = (plane, row, slot), 16 bytes = 8 bf16; the three planes of a row are 16 bf16 apart
    const __bf16* const w3 = reinterpret_cast<const __bf16*>(p.w3);
    unsigned b_off[kBIters]; int b_lds[kBIters]; bool b_ok[kBIters];
#pragma unroll
    for (int i = 0; i < kBIters; ++i) {
        const int u = tid + i * NT;
        const int pl = u / (BN * 2), within = u % (BN * 2), row = within >> 1, slot = within & 1;
        b_ok[i] = u < kBUnits && n0 + row < p.Np;
        b_off[i] = (unsigned)(((n0 + row) * 3 + pl) * 16 + slot * 8);
        b_lds[i] = (pl * BN + row) * 2 + (slot ^ ((row >> 3) & 1));
    }
    const long w_tile = 3l * p.Np * 16;                                    // bf16 elements per (chunk, tap)

    float4 a_reg[kAIters];
    // PF = 1: two register sets, loaded by global_load_dwordx4 written as inline asm -- hidden from hipcc's wait-count insertion like the
    // weight copies (with ordinary loads it put s_waitcnt vmcnt(0) in front of every use: one K-tile in flight again) -- and tied to
    // their consumer by an asm s_waitcnt that takes the registers as in/out operands (X3_TIE_WAIT): nothing reads them before it.
#if defined(DR_EMU)
    float4 a_hid[PF == 1 ? kAIters : 1], a_hid2[PF == 1 ? kAIters : 1];
#else
    dr_f32x4 a_hid[PF == 1 ? kAIters : 1], a_hid2[PF == 1 ? kAIters : 1];
#endif
    float4 b_reg0, b_reg1, b_reg2;                                          // (scalars: hipcc keeps a float4[3] refilled inside the unrolled K loop in scratch)
    static_assert(BD || kBIters <= 3, "weight loader mapping");
    int ld_kc = 0, ld_dy = -pad, ld_dx = -pad, ld_tap = 0;
    const float* ld_x = p.x + (long)(ld_dy * p.W + ld_dx) * p.x_cs;
    const __bf16* ld_w = w3;
    int a_nv[kAIters], a_nv2[PF ? kAIters : 1];
    auto load_tile_to = [&](auto& a_reg, int (&a_nv)[kAIters]) __attribute__((always_inline)) {
        const bool tail = ld_kc + CK > p.Cin;                              // uniform: this chunk crosses Cin
#pragma unroll
        for (int i = 0; i < kAIters; ++i) {
            bool ok = (a_taps[i] >> ld_tap) & 1u;
            int nv = 4;
            if (tail) {
                const int left = p.Cin - (ld_kc + a_q[i] * 4);
                nv = left < 0 ? 0 : (left > 4 ? 4 : left);
                ok = ok && nv > 0;
            }
#if !defined(DR_EMU)
            if constexpr (PF == 1) {
                const float* src = ok ? ld_x + ld_kc + a_off[i] : p.zeros;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a_reg[i]) : "v"(src) : "memory");
            } else
#endif
            a_reg[i] = *reinterpret_cast<const float4*>(ok ? ld_x + ld_kc + a_off[i] : p.zeros);
            a_nv[i] = ok ? nv : 4;
        }
#define X3_LOAD_B(i) *reinterpret_cast<const float4*>(b_ok[i] ? reinterpret_cast<const void*>(ld_w + b_off[i]) : reinterpret_cast<const void*>(p.zeros))
        if constexpr (RING != 2 && !BD) {                                  // (RING = 2 / BD copy the weight planes by LDS-DMA: dma_b / dma_bd below)
            b_reg0 = X3_LOAD_B(0);
            if constexpr (kBIters > 1) b_reg1 = X3_LOAD_B(1);
            if constexpr (kBIters > 2) b_reg2 = X3_LOAD_B(2);
        }
#undef X3_LOAD_B
        ++ld_tap;
        if (++ld_dx > pad) { ld_dx = -pad; ++ld_dy; }
        if (ld_tap == taps) {
            ld_tap = 0;
            ld_dy = ld_dx = -pad;
            ld_kc += CK;
        }
        ld_x = p.x + (long)(ld_dy * p.W + ld_dx) * p.x_cs;
        ld_w += w_tile;
    };
    auto load_tile = [&]() __attribute__((always_inline)) { load_tile_to(a_reg, a_nv); };
    const bool ragged = (p.Cin & 3) != 0;
    auto store_tile_from = [&](const int buf, const bool was_tail, const auto& a_reg, const int (&a_nv)[kAIters]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < kAIters; ++i) {
            const int r = a_row[i], q = a_q[i];
            if (BM * 4 < NT * kAIters && r >= BM) continue;
            float4 v;
            v = x3_as_f4(a_reg[i]);
            if (ragged && was_tail) {
                const int nv = a_nv[i];
                v.y = nv > 1 ? v.y : 0.f;
                v.z = nv > 2 ? v.z : 0.f;
                v.w = nv > 3 ? v.w : 0.f;
            }
            uint2 h0, h1, h2;
            x3_split4(v, h0, h1, h2);
            const int slot = (q >> 1) ^ ((r >> 3) & 1);
            uint2* d0 = reinterpret_cast<uint2*>(&DR_AS(buf)[0][r][slot]) + (q & 1);
            uint2* d1 = reinterpret_cast<uint2*>(&DR_AS(buf)[1][r][slot]) + (q & 1);
            uint2* d2 = reinterpret_cast<uint2*>(&DR_AS(buf)[2][r][slot]) + (q & 1);
            *d0 = h0; *d1 = h1; *d2 = h2;
        }
        if constexpr (RING != 2 && !BD) {
            float4* const bs = &DR_BS(buf)[0][0][0];
            if (kBUnits % NT == 0 || tid < kBUnits) bs[b_lds[0]] = b_reg0;
            if constexpr (kBIters > 1) { if (kBUnits % NT == 0 || tid + NT < kBUnits) bs[b_lds[1]] = b_reg1; }
            if constexpr (kBIters > 2) { if (kBUnits % NT == 0 || tid + 2 * NT < kBUnits) bs[b_lds[2]] = b_reg2; }
        }
    };
    auto store_tile = [&](const int buf, const bool was_tail) __attribute__((always_inline)) { store_tile_from(buf, was_tail, a_reg, a_nv); };
    // RING: the weight planes of a K-tile go HBM / L2 -> LDS without registers (global_load_lds_dwordx4): wave w copies the 64-unit
    // chunks w, w + 4, ... of the tile's 3 * BN * 2 units; the LDS image is lane-linear, so the slot swizzle moves to the SOURCE
    // address (unit L = (plane, row, physical slot) fetches logical slot = physical ^ ((row >> 3) & 1)).
    unsigned bd_off[kBIters]; bool bd_ok[kBIters];
#pragma unroll
    for (int i = 0; i < kBIters; ++i) {
        const int L = (wave + NW * i) * 64 + lane;
        const int pl = L / (BN * 2), within = L % (BN * 2), row = within >> 1, slot = (within & 1) ^ ((row >> 3) & 1);
        bd_ok[i] = L < kBUnits && n0 + row < p.Np;
        bd_off[i] = (unsigned)(((n0 + row) * 3 + pl) * 16 + slot * 8);
    }
    const __bf16* dma_w = w3;
    auto dma_b = [&](const int st) __attribute__((always_inline)) {
        if constexpr (RING != 2) return;
#pragma unroll
        for (int i = 0; i < kBIters; ++i)
            if ((wave + NW * i) * 64 < kBUnits)                             // (wave-uniform)
                dr_glds16(bd_ok[i] ? reinterpret_cast<const float*>(dma_w + bd_off[i]) : p.zeros,
                          reinterpret_cast<float*>(&DR_BS(st)[0][0][0] + (wave + NW * i) * 64));
        dma_w += w_tile;
    };

    // BD: wave w issues the 1 KB copies 3 w .. 3 w + 2 of a stage's kBUnits / 64 (lane L of copy q = unit 64 q + L = (plane, row, physical
    // slot), fetching the logical slot; rows beyond Np read zeros: x3_dma.h)
    constexpr int kBInstr = kBUnits / 64;
    constexpr int kBQ = (kBInstr + NW - 1) / NW > 3 ? (kBInstr + NW - 1) / NW : 3;      // copies per wave (three; four for the 160-column tile's 15 over four waves)
    static_assert(!BD || (kBUnits % 64 == 0 && kBInstr <= kBQ * NW), "weight copies");
    const P3Src srcB = p3_src(p.w3, 0, BD ? (size_t)T_total * p.Np * 96 : 0);
    unsigned bq_voff[kBQ];
    const int bq_n = !BD ? 0 : kBInstr - wave * kBQ < 0 ? 0 : (kBInstr - wave * kBQ > kBQ ? kBQ : kBInstr - wave * kBQ);
#pragma unroll
    for (int j = 0; j < kBQ; ++j) {
        const int u = (wave * kBQ + j) * 64 + lane;
        const int pl = u / (BN * 2), within = u % (BN * 2), row = within >> 1, ls = (within & 1) ^ ((row >> 3) & 1);
        bq_voff[j] = (BD && u < kBUnits && n0 + row < p.Np) ? (unsigned)(((n0 + row) * 3 + pl) * 32 + ls * 16) : kP3Oob;
    }
    unsigned bq_soff = 0;
    auto dma_bd = [&](const int st) __attribute__((always_inline)) {
        if constexpr (!BD) return;
#pragma unroll
        for (int j = 0; j < kBQ; ++j)
            if (j < bq_n) p3_dma16(srcB, bq_voff[j], bq_soff, reinterpret_cast<unsigned char*>(&DR_BS(st)[0][0][0]), (unsigned)((wave * kBQ + j) * 1024));
        bq_soff += (unsigned)p.Np * 96u;
    };

    // Two accumulators per output tile: `acc` takes the leading products a0*b0, `lo` the five correction products (each <= 2^-8 of
    // the leading one).  Added to ONE accumulator every correction would round at the magnitude of the running sum -- six roundings
    // per k instead of one, which measured 4-7x the fp32 kernel's error on operands spread over many binades; in its own accumulator
    // the corrections round at 2^-8 of that magnitude and join the sum once, after the K loop.
    using AccT = dr_f32x16;
    constexpr int NR = 16;
    AccT acc[kTM][kTN], lo[LO ? kTM : 1][LO ? kTN : 1];
#pragma unroll
    for (int i = 0; i < kTM; ++i)
#pragma unroll
        for (int j = 0; j < kTN; ++j)
#pragma unroll
            for (int r = 0; r < NR; ++r) { acc[i][j][r] = 0.f; if constexpr (LO) lo[i][j][r] = 0.f; }

    const int lk = lane >> 5;
    const int li = lane & 31;
    const int fslot = lk ^ ((li >> 3) & 1);                                // this lane's LDS slot of every row it reads (rows = 32*t + li)
    auto& LOACC = *reinterpret_cast<AccT(*)[kTM][kTN]>(LO ? &lo[0][0] : &acc[0][0]);
#define X3_READ_A(d, pl, st) _Pragma("unroll") for (int i = 0; i < kTM; ++i) d[i] = DR_AS(st)[pl][wm * kWTM + i * 32 + li][fslot]
#define X3_READ_B(d, pl, st) _Pragma("unroll") for (int j = 0; j < kTN; ++j) d[j] = DR_BS(st)[pl][wn * kWTN + j * 32 + li][fslot]
#define X3_MMA(c, a, b)                                                                                                          \
    _Pragma("unroll") for (int i = 0; i < kTM; ++i) _Pragma("unroll") for (int j = 0; j < kTN; ++j)                              \
        c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dr_bf16x8, a[i]), __builtin_bit_cast(dr_bf16x8, b[j]), c[i][j], 0, 0, 0)
    const bool tail0 = CK > p.Cin;
    if constexpr (RING) {
        // ---- three stages: tiles 0 and 1 in LDS, tile 2's weights in LDS and its pixels in the staging registers, the planes 0 of
        // tile 0 in a0 / b0.  Steady state of tile t (stage cur = t % 3):
        //   read planes 2 of t | MFMA a0*b0, a2*b0, a0*b2 | read planes 1 of t | split + ds_write the pixels of t+2 (loaded after the
        //   previous barrier) | wait for the weight copy of t+2 (issued after the previous barrier) | BARRIER | issue the pixel loads and
        //   the weight copy of t+3 (into stage cur: every read of it was issued before the barrier) | MFMA a1*b0 | read b0 of t+1 |
        //   MFMA a0*b1 | read a0 of t+1 | MFMA a1*b1
        load_tile(); dma_b(0);
        store_tile(0, tail0);
        bool pend_tail = false;                                             // of the tile held in the staging registers
        if (T_total > 1) { const bool tl = ld_kc + CK > p.Cin; load_tile(); dma_b(1); store_tile(1, tl); }
        if (T_total > 2) { pend_tail = ld_kc + CK > p.Cin; load_tile(); dma_b(2); }
#if !defined(DR_EMU)
        if constexpr (RING == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA copies have landed (a barrier alone does not say so)
#endif
        __syncthreads();
        float4 a0[kTM], b0[kTN];
        X3_READ_A(a0, 0, 0); X3_READ_B(b0, 0, 0);
        // one K-tile: tile t in stage `cur`; more1 / more2 / more3: tiles t+1 / t+2 / t+3 exist
        auto ring_tile = [&](const int cur, const int nxt, const int st, const bool more1, const bool more2, const bool more3) __attribute__((always_inline)) {
            float4 ax[kTM], bx[kTN];
            X3_READ_A(ax, 2, cur); X3_READ_B(bx, 2, cur);
            X3_MMA(acc, a0, b0);                                           // starts at once: its operands were fetched under the previous tile
            X3_MMA(LOACC, ax, b0);                                         // a2*b0
            X3_MMA(LOACC, a0, bx);                                         // a0*b2
            X3_READ_A(ax, 1, cur); X3_READ_B(bx, 1, cur);
            if (more2) store_tile(st, pend_tail);                          // the pixels of tile t+2
#if !defined(DR_EMU)
            if constexpr (RING == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the weight copy of tile t+2 (nothing else is outstanding here)
#endif
            __syncthreads();                                                // (under the twelve MFMAs queued above)
            if (more3) { pend_tail = ld_kc + CK > p.Cin; load_tile(); dma_b(cur); }
            X3_MMA(LOACC, ax, b0);                                         // a1*b0: the last reader of b0
            if (more1) { X3_READ_B(b0, 0, nxt); }
            X3_MMA(LOACC, a0, bx);                                         // a0*b1: the last reader of a0
            if (more1) { X3_READ_A(a0, 0, nxt); }
            X3_MMA(LOACC, ax, bx);                                         // a1*b1
        };
        const int T3 = T_total - T_total % 3;
        for (int t = 0; t < T3; t += 3) {
            ring_tile(0, 1, 2, true, true, t + 3 < T_total);
            ring_tile(1, 2, 0, true, t + 3 < T_total, t + 4 < T_total);
            ring_tile(2, 0, 1, t + 3 < T_total, t + 4 < T_total, t + 5 < T_total);
        }
        if (T_total - T3 == 2) {
            ring_tile(0, 1, 2, true, false, false);
            ring_tile(1, 2, 0, false, false, false);
        } else if (T_total - T3 == 1) {
            ring_tile(0, 1, 2, false, false, false);
        }
        __syncthreads();                                                    // (the epilogue reuses stage 0 as scratch)
    } else if constexpr (PF == 1) {
    // ---- two stages, the pixels of TWO K-tiles in flight (the short-K 1x1 layers: eight K-tiles behind one prologue; a K-tile's MFMAs
    // last a third of an HBM round trip).  Tile t (stage t & 1): weight copy of t+1 | pixel loads of t+2 into register set t & 1 (free:
    // tile t's pixels went to LDS a tile ago) | the MFMAs of t | split + ds_write the pixels of t+1 from the other set | wait until only
    // the loads of t+2 are outstanding (loads return in order: the copy of t+1, issued BEFORE them, has landed) | barrier.
#if defined(DR_EMU)
#define X3_WAIT_BUT_PIXELS() ((void)0)
#define X3_TIE_WAIT(regs, keep) ((void)0)
#else
#define X3_WAIT_BUT_PIXELS() do { if constexpr (kAIters == 1) P3_WAIT_VM(1); else if constexpr (kAIters == 2) P3_WAIT_VM(2); else P3_WAIT_VM(0); } while (0)
    // "at most `keep` x kAIters newer loads outstanding" AND the registers of `regs` are the wait's operands: their readers cannot be
    // scheduled in front of it (the loads that fill them are invisible to the compiler)
#define X3_TIE_WAIT(regs, keep)                                                                                                   \
    do {                                                                                                                          \
        _Pragma("unroll") for (int i_ = 0; i_ < kAIters; ++i_) {                                                                  \
            if ((keep) && kAIters == 1) asm volatile("s_waitcnt vmcnt(1)" : "+v"(regs[i_])::"memory");                            \
            else if ((keep) && kAIters == 2) asm volatile("s_waitcnt vmcnt(2)" : "+v"(regs[i_])::"memory");                       \
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(regs[i_])::"memory");                                                   \
        }                                                                                                                         \
    } while (0)
#endif
    static_assert(kAIters <= 2, "hand-counted waits");
    dma_bd(0);
    load_tile_to(a_hid, a_nv);
    bool tl2 = false;
    if (T_total > 1) { tl2 = ld_kc + CK > p.Cin; load_tile_to(a_hid2, a_nv2); }
    X3_TIE_WAIT(a_hid, T_total > 1);                                      // the weight copy and the pixels of tile 0 (tile 1's stay in flight)
    store_tile_from(0, tail0, a_hid, a_nv);
    if (T_total > 1) X3_WAIT_BUT_PIXELS(); else P3_WAIT_VM(0);            // (its lgkmcnt(0): the ds_writes)
    __builtin_amdgcn_s_barrier();
    bool tl1 = false;
    auto pf_tile = [&](const int buf, const bool more1, const bool more2, auto& rl, int (&nvl)[kAIters], bool& tl_l,
                       auto& rs, const int (&nvs)[kAIters], const bool tl_s) __attribute__((always_inline)) {
        if (more1) dma_bd(buf ^ 1);
        if (more2) { tl_l = ld_kc + CK > p.Cin; load_tile_to(rl, nvl); }
        float4 a0[kTM], b0[kTN], ax[kTM], bx[kTN];
        X3_READ_A(a0, 0, buf); X3_READ_B(b0, 0, buf); X3_READ_A(ax, 2, buf); X3_READ_B(bx, 2, buf);
        X3_MMA(LOACC, ax, b0);
        X3_MMA(LOACC, a0, bx);
        X3_READ_A(ax, 1, buf); X3_READ_B(bx, 1, buf);
        X3_MMA(acc, a0, b0);
        X3_MMA(LOACC, ax, b0);
        X3_MMA(LOACC, a0, bx);
        X3_MMA(LOACC, ax, bx);
        __builtin_amdgcn_sched_barrier(0);
        if (more1) { X3_TIE_WAIT(rs, more2); store_tile_from(buf ^ 1, tl_s, rs, nvs); }   // (in order: the weight copy of t+1, issued before the loads of t+2, has landed too)
        __builtin_amdgcn_sched_barrier(0);
        if (more2) X3_WAIT_BUT_PIXELS(); else P3_WAIT_VM(0);
        __builtin_amdgcn_s_barrier();
    };
    const int T_pairs = T_total & ~1;
    for (int t = 0; t < T_pairs; t += 2) {
        pf_tile(0, true, t + 2 < T_total, a_hid, a_nv, tl1, a_hid2, a_nv2, tl2);
        pf_tile(1, t + 2 < T_total, t + 3 < T_total, a_hid2, a_nv2, tl2, a_hid, a_nv, tl1);
    }
    if (T_total & 1) pf_tile(0, false, false, a_hid, a_nv, tl1, a_hid2, a_nv2, tl2);
#undef X3_WAIT_BUT_PIXELS
#undef X3_TIE_WAIT
    } else {
    load_tile(); dma_bd(0);
    store_tile(0, tail0);
    if constexpr (BD) { P3_WAIT_VM(0); __builtin_amdgcn_s_barrier(); } else __syncthreads();

    auto k_tile = [&](const int buf, const bool more) __attribute__((always_inline)) {
        const bool was_tail = ld_kc + CK > p.Cin;                          // of the tile being fetched now
        if (more) { if constexpr (!kNoA) load_tile(); if constexpr (!kNoB) dma_bd(buf ^ 1); }
        // fragments are read plane by plane, the planes 2 first: their registers are reused by the planes 1 (eight fragments live, not twelve)
        float4 a0[kTM], b0[kTN], ax[kTM], bx[kTN];
        X3_READ_A(a0, 0, buf); X3_READ_B(b0, 0, buf); X3_READ_A(ax, 2, buf); X3_READ_B(bx, 2, buf);
        X3_MMA(LOACC, ax, b0);                                                // a2*b0
        X3_MMA(LOACC, a0, bx);                                                // a0*b2
        X3_READ_A(ax, 1, buf); X3_READ_B(bx, 1, buf);
        X3_MMA(acc, a0, b0);                                               // the leading products run while the planes 1 arrive
        X3_MMA(LOACC, ax, b0);                                                // a1*b0   (the corrections in the ring's order: same bits)
        X3_MMA(LOACC, a0, bx);                                                // a0*b1
        X3_MMA(LOACC, ax, bx);                                                // a1*b1
        if constexpr (PF == 2) __builtin_amdgcn_sched_barrier(0);
        if constexpr (!kNoA) { if (more) store_tile(buf ^ 1, was_tail); }
        if constexpr (kNoSync) {
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (BD) {                                          // the tile's MFMAs are issued, then: the copy has landed, every LDS access returned
            __builtin_amdgcn_sched_barrier(0);
            P3_WAIT_VM(0);
            __builtin_amdgcn_s_barrier();
        } else __syncthreads();
    };
    const int T_pairs = T_total & ~1;
    for (int t = 0; t < T_pairs; t += 2) {
        k_tile(0, true);
        k_tile(1, t + 2 < T_total);
    }
    if (T_total & 1) k_tile(0, false);
    }
#undef X3_READ_A
#undef X3_READ_B
#undef X3_MMA
#pragma unroll
    for (int i = 0; i < kTM; ++i)
#pragma unroll
        for (int j = 0; j < kTN; ++j) { if constexpr (LO) acc[i][j] += lo[i][j]; }

    // ---- epilogue: conv_epilogue.inc (the fp32 copy) -------------------------------------------------------------------------------
    double s1[kTN], s2[kTN];
#pragma unroll
    for (int j = 0; j < kTN; ++j) s1[j] = s2[j] = 0.0;
    constexpr int EP_TM = kTM, EP_TN = kTN;
    const int ep_m0 = m0 + wm * kWTM, ep_n0 = n0 + wn * kWTN;
    const unsigned ep_rows = 0xFFFFu;
#if !defined(DR_X3_EPB)
#define DR_X3_EPB 4                                          // (A/B builds: 8)
#endif
    constexpr int EP_BATCH_ROWS = x3_waves_per_simd<BM, BN, LO, NW>() == 4 ? DR_X3_EPB : 8;  // (four waves per SIMD live on 128 registers: four rows of epilogue loads in flight)
    constexpr int EP_TS = MF, EP_NR = NR;
    const int ep_lg = lk, ep_lc = li;
    {
        constexpr bool EP_Y16 = false, EP_B16 = false, EP_B16_CONST = false;
#include "conv_epilogue.inc"
    }
    if (p.stat_part) {
        double* red = reinterpret_cast<double*>(&As0[0][0][0]);            // the operand tiles are dead: the K loop ended on a barrier
        static_assert(sizeof(As0) >= sizeof(double) * 2 * WM * BN, "stat scratch does not fit the operand tile");
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
        // one partial row per 128 pixel rows whatever BM (BM = 256: two rows per workgroup, each the sum of its two 64-row waves -- the
        // rows, and the bits, of the 128-row kernel)
        constexpr int kHalves = BM / 128, kWH = WM / kHalves;
        static_assert(BM % 128 == 0 && WM % kHalves == 0, "statistics rows are 128 pixel rows");
        const int srows = (M + 127) >> 7;
        for (int e = tid; e < 2 * BN * kHalves; e += NT) {
            const int half = e / (2 * BN), which = (e / BN) % 2, col = e % BN, n = n0 + col;
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < kWH; ++w) t += red[(which * WM + half * kWH + w) * BN + col];
            const int srow = mblk * kHalves + half;
            if (n < p.Cout && srow < srows) p.stat_part[((long)which * p.Cout + n) * srows + srow] = t;
        }
    }
    (void)wk; (void)WK;
}

#undef DR_AS
#undef DR_BS

}  // namespace dr
