// conv_igemm2.h -- second-generation implicit-GEMM conv: A operand straight from HBM/L2 to registers.
//
// Same contract as conv_igemm.h (ConvParams, fused epilogue).  What changed and why (round-1 ablations on
// MI355X, 3x3 256->256 @ B=40: product kernel 584 us; without global->LDS refills 459 us; the refill
// side alone 186 us): the activation tile's trip through LDS (8 transposing ds_write_b32 per thread and
// K-tile plus their address VALU) buys nothing -- with the fp32 32x32x2 MFMA each lane consumes ONE
// pixel's channels, so the lane can load them itself:
//
//   lane (i = l&31, h = l>>5) owns pixel row i of its wave tile and the 8 channels  kc+8h .. kc+8h+7
//   of every 16-channel K-tile (two float4 loads, 32 contiguous bytes);  MFMA step j of the tile
//   contracts the channel pair {kc+j (lanes 0-31), kc+8+j (lanes 32-63)}.
//
// The reduction order inside a K-tile is therefore (0,8),(1,9),..,(7,15) instead of (0,1),(2,3),..:
// still one exact k-ordered fp32 fma chain per output, fixed for a given shape (bit-reproducible,
// batch-invariant).  Only the weight tile Bs[16][BN] goes through LDS (shared by the 4 waves).
// Zero-padding of SAME convs, ragged channel tails and the depth row-mask are per-lane predicates.
#pragma once
#include "conv_igemm.h"

namespace dr {

template <int BM, int BN, int WM, int WN, int ABL = 0>
__global__ __launch_bounds__(256) void conv_igemm2_kernel(const ConvParams p) {
    constexpr int BK = 16;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int BITERS = (BK * (BN / 4) + 255) / 256;
    static_assert(WM * WN == 4 && WTM % 32 == 0 && WTN % 32 == 0, "4 waves, 32x32 MFMA tiles");
    __shared__ float Bs[2][BK][BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int HW = p.H * p.W;
    const int M = p.B * HW;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int taps = p.ksize * p.ksize;
    const int KT = p.Kp / BK;
    const int T_total = taps * KT;
    const int pad = p.ksize / 2;

    // ---- per-lane pixel rows ---------------------------------------------------------------------
    int r_y[TM], r_x[TM];
    long r_base[TM];                 // element offset of (pixel, channel 8h) or -1
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WTM + i * 32 + li;
        bool ok = m < M;
        if (ok && p.rowmask) ok = !(p.rowmask[m] < p.mask_thresh);
        const int rem = ok ? (m % HW) : 0;
        r_y[i] = rem / p.W;
        r_x[i] = rem % p.W;
        r_base[i] = ok ? (long)m * p.x_cs + p.x_coff + 8 * lh : -1;
    }

    float4 a_nxt[TM][2];
    float4 b_reg[BITERS];
    auto load_tile = [&](int t) {
        const int tap = t / KT;
        const int kc = (t - tap * KT) * BK;
        const int dy = tap / p.ksize - pad, dx = tap % p.ksize - pad;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int yy = r_y[i] + dy, xx = r_x[i] + dx;
            const bool row_ok = r_base[i] >= 0 && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            const float* src = p.x + r_base[i] + (long)(dy * p.W + dx) * p.x_cs + kc;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                const int c = kc + 8 * lh + 4 * q;
                if (row_ok && c < p.Cin) {
                    if (c + 4 <= p.Cin) {
                        v = *reinterpret_cast<const float4*>(src + 4 * q);
                    } else {                    // ragged channel tail (Cin % 4 != 0)
                        v.x = src[4 * q];
                        if (c + 1 < p.Cin) v.y = src[4 * q + 1];
                        if (c + 2 < p.Cin) v.z = src[4 * q + 2];
                    }
                }
                a_nxt[i][q] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < BITERS; ++i) {
            const int idx = tid + i * 256;
            const int krow = idx / (BN / 4), n4 = idx % (BN / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (krow < BK && n0 + n4 * 4 < p.Np)
                v = *reinterpret_cast<const float4*>(p.w + ((long)tap * p.Kp + kc + krow) * p.Np + n0 + n4 * 4);
            b_reg[i] = v;
        }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int i = 0; i < BITERS; ++i) {
            const int idx = tid + i * 256;
            const int krow = idx / (BN / 4), n4 = idx % (BN / 4);
            if (krow < BK) *reinterpret_cast<float4*>(&Bs[buf][krow][n4 * 4]) = b_reg[i];
        }
    };

    dr_f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float a_cur[TM][8];
    load_tile(0);
    store_b(0);
    __syncthreads();

    for (int t = 0; t < T_total; ++t) {
        const int buf = t & 1;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            a_cur[i][0] = a_nxt[i][0].x; a_cur[i][1] = a_nxt[i][0].y; a_cur[i][2] = a_nxt[i][0].z; a_cur[i][3] = a_nxt[i][0].w;
            a_cur[i][4] = a_nxt[i][1].x; a_cur[i][5] = a_nxt[i][1].y; a_cur[i][6] = a_nxt[i][1].z; a_cur[i][7] = a_nxt[i][1].w;
        }
        if (ABL != 1 && t + 1 < T_total) load_tile(t + 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float b[TN];
#pragma unroll
            for (int n = 0; n < TN; ++n) b[n] = Bs[buf][8 * lh + j][wn * WTN + n * 32 + li];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    if (ABL == 2) acc[i][n][0] = fmaf(a_cur[i][j], b[n], acc[i][n][0]);
                    else acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i][j], b[n], acc[i][n], 0, 0, 0);
                }
        }
        if (ABL != 1 && t + 1 < T_total) store_b(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue (identical to conv_igemm.h) --------------------------------------------------------
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WTN + j * 32 + li;
        const bool n_ok = n < p.Cout;
        const float sc = (n_ok && p.scale) ? p.scale[n] : 1.f;
        const float sh = (n_ok && p.shift) ? p.shift[n] : 0.f;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (ABL == 3 && acc[i][j][r] != 12345.678f) continue;
                if (m < M && n_ok && !(p.out_rowmask && p.out_rowmask[m] < p.out_mask_thresh)) {
                    const float raw = acc[i][j][r];
                    s1 += (double)raw;
                    s2 += (double)raw * (double)raw;
                    float v = raw * sc + sh;
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.drop) v = p.drop[(long)m * p.Cout + n] ? v * 2.f : 0.f;
                    else if (p.drop_rng) v = dropout_keep(p.drop_seed, (unsigned long long)m * p.Cout + n) ? v * 2.f : 0.f;
                    if (p.res) v += p.res[(long)m * p.res_cs + p.res_coff + n];
                    p.y[(long)m * p.y_cs + p.y_coff + n] = v;
                }
            }
        }
        if (p.stat_sum) {
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (lh == 0 && n_ok) {
                atomicAdd(&p.stat_sum[n], s1);
                atomicAdd(&p.stat_sq[n], s2);
            }
        }
    }
}

}  // namespace dr
