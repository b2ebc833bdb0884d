// conv_igemm.h -- implicit-GEMM stride-1 SAME convolution (1x1 / 3x3) on the fp32 matrix cores.
//
// Replaces tf.nn.conv2d + BatchReNorm/bias + ReLU (+ residual add, + dropout) as composed by
// network/slim/ops.py:219-299 and network/um_v1.py:18-48 (reference, NHWC x HWIO, fp32).
//
//   GEMM view:  M = B*H*W output pixels,  N = Cout,  K = taps*Cin
//   A[m][k]   = x[b, y+dy, x+dx, c]   gathered on the fly (zero outside the image: TF 'SAME')
//   B[k][n]   = packed weights [tap][Kp][Np]  (HWIO with Cin->Kp, Cout->Np zero padding)
//   D         = v_mfma_f32_32x32x2_f32 chains: exact fp32 (one rounding per product, k-ordered),
//               157 TFLOP/s peak on gfx950 -- the roofline this kernel is priced against.
//
// Block = 256 threads = 4 waves; wave tile = (BM/WM) x (BN/WN) made of 32x32 MFMA tiles.
// LDS holds As[BK][BM(+pad)] (k-major, so a lane's A fragment A[i=l&31][k=l>>5] is a
// conflict-free row read) and Bs[BK][BN]; both are double buffered with register prefetch of the
// next K-tile, one barrier per K-tile.
//
// Epilogue (per lane = one output channel, 16 rows):  v = acc*scale[n] + shift[n]; relu;
// dropout keep mask (x2); + residual;  optional per-channel sum / sum-of-squares of the RAW
// accumulator for train-mode BatchReNorm (tf.nn.moments, ops.py:132) via fp64 atomics.
#pragma once
#include "dr_platform.h"

namespace dr {

struct ConvParams {
    const float* x; int x_cs; int x_coff; int Cin;
    int B, H, W;
    int ksize;                         // 1 or 3
    const float* w; int Kp; int Np;
    float* y; int y_cs; int y_coff; int Cout;
    const float* scale;                // nullable: 1
    const float* shift;                // nullable: 0
    int relu;
    const float* res; int res_cs; int res_coff;      // nullable
    const float* rowmask; float mask_thresh;         // nullable: A row zeroed where rowmask[m] < thresh
    const unsigned char* drop;                       // nullable: keep mask [M][Cout], kept -> x2
    int drop_rng; unsigned long long drop_seed;      // drop_rng != 0: counter-based keep bit instead of `drop`
    const float* out_rowmask; float out_mask_thresh; // nullable: rows with out_rowmask[m] < thresh are not written
    double* stat_sum; double* stat_sq;               // nullable: per-channel moments of acc
};

// stateless keep bit for dropout(0.5): splitmix64 finaliser of (seed, element index)
__device__ __forceinline__ bool dropout_keep(unsigned long long seed, unsigned long long idx) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (z >> 17) & 1ull;
}

template <int BM, int BN, int WM, int WN, int BK_ = 16>
struct ConvTile {
    static constexpr int kBK = BK_;
    static constexpr int kThreads = 256;
    // As row stride: the transposing ds_write_b32 of (k4, m) lane pairs is conflict-free when
    // 4*kSA*k4 mod 32 spreads over distinct bank groups: kSA = 2 (mod 8) for BK=16, odd for BK=32
    static constexpr int kSA = BM + (BK_ == 16 ? 2 : 1);
    static constexpr int kSB = BN;
    static constexpr int kWTM = BM / WM;          // wave tile rows
    static constexpr int kWTN = BN / WN;
    static constexpr int kTM = kWTM / 32;
    static constexpr int kTN = kWTN / 32;
    static constexpr int kAIters = (BM * (kBK / 4)) / kThreads;
    static constexpr int kBIters = (kBK * (BN / 4) + kThreads - 1) / kThreads;
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(kWTM % 32 == 0 && kWTN % 32 == 0, "wave tile must be made of 32x32 MFMA tiles");
    static_assert((BM * (kBK / 4)) % kThreads == 0, "A loader mapping");
    static constexpr size_t kLdsBytes = size_t(2) * kBK * (kSA + kSB) * sizeof(float);
};

// ABL (profiling ablations, product code uses 0): 1 = no global->LDS refills after the first K-tile,
// 2 = MFMA replaced by one VALU fma per fragment pair, 3 = no epilogue stores.
template <int BM, int BN, int WM, int WN, int ABL = 0, int BK_ = 16>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
    using T = ConvTile<BM, BN, WM, WN, BK_>;
    constexpr int BK = T::kBK;
    constexpr int SA = T::kSA;
    constexpr int SB = T::kSB;
    __shared__ float As[2][BK][SA];
    __shared__ float Bs[2][BK][SB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int HW = p.H * p.W;
    const int M = p.B * HW;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int taps = p.ksize * p.ksize;
    const int KT = p.Kp / BK;
    const int T_total = taps * KT;
    const int pad = p.ksize / 2;

    // ---- per-thread loader bookkeeping (same rows for every K-tile) ------------------------------
    // Everything lane-dependent is computed ONCE: a 32-bit element offset, a 9-bit "tap stays inside the
    // image" mask (TF 'SAME' zero padding + the optional depth row-mask) and the LDS coordinates.  Per
    // K-tile only the wave-uniform part moves (tap shift and channel chunk, scalar registers), so a tile
    // refill costs a handful of VALU ops per load instead of 64-bit address math and nested predicates.
    int a_row[T::kAIters], a_k4[T::kAIters];
    unsigned a_off[T::kAIters];       // element offset of (pixel m, channel 4*k4) relative to p.x
    unsigned a_taps[T::kAIters];      // bit t: tap t reads a valid pixel for this row
#pragma unroll
    for (int i = 0; i < T::kAIters; ++i) {
        const int idx = tid + i * T::kThreads;
        const int row = idx / (BK / 4);
        a_row[i] = row;
        a_k4[i] = idx % (BK / 4);
        const int m = m0 + row;
        bool ok = m < M;
        if (ok && p.rowmask) ok = !(p.rowmask[m] < p.mask_thresh);
        const int rem = ok ? (m % HW) : 0;
        const int y = rem / p.W, x = rem % p.W;
        unsigned mask = 0;
        if (ok) {
            for (int t = 0; t < taps; ++t) {
                const int yy = y + t / p.ksize - pad, xx = x + t % p.ksize - pad;
                if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) mask |= 1u << t;
            }
        }
        a_taps[i] = mask;
        a_off[i] = ok ? (unsigned)((long)m * p.x_cs + p.x_coff + a_k4[i] * 4) : 0u;
    }
    unsigned b_off[T::kBIters];
    bool b_ok[T::kBIters];
#pragma unroll
    for (int i = 0; i < T::kBIters; ++i) {
        const int idx = tid + i * T::kThreads;
        const int krow = idx / (BN / 4), n4 = idx % (BN / 4);
        b_ok[i] = krow < BK && n0 + n4 * 4 < p.Np;
        b_off[i] = (unsigned)(krow * p.Np + n0 + n4 * 4);
    }

    float4 a_reg[T::kAIters];
    float4 b_reg[T::kBIters];

    auto load_tile = [&](int t) {
        const int tap = t / KT;                               // wave-uniform
        const int kc = (t - tap * KT) * BK;
        const int dy = tap / p.ksize - pad, dx = tap % p.ksize - pad;
        const float* xbase = p.x + ((long)(dy * p.W + dx) * p.x_cs + kc);       // uniform: SGPR pair
        const float* wbase = p.w + ((long)tap * p.Kp + kc) * p.Np;
        const bool whole = kc + BK <= p.Cin;                  // uniform: no channel predicate needed
#pragma unroll
        for (int i = 0; i < T::kAIters; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((a_taps[i] >> tap) & 1u) {
                const float* src = xbase + a_off[i];
                if (whole) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    const int c = kc + a_k4[i] * 4;
                    if (c + 4 <= p.Cin) {
                        v = *reinterpret_cast<const float4*>(src);
                    } else if (c < p.Cin) {                   // ragged channel tail (Cin % 4 != 0)
                        v.x = src[0];
                        if (c + 1 < p.Cin) v.y = src[1];
                        if (c + 2 < p.Cin) v.z = src[2];
                    }
                }
            }
            a_reg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < T::kBIters; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b_ok[i]) v = *reinterpret_cast<const float4*>(wbase + b_off[i]);
            b_reg[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < T::kAIters; ++i) {
            const int k = a_k4[i] * 4, r = a_row[i];
            As[buf][k + 0][r] = a_reg[i].x;
            As[buf][k + 1][r] = a_reg[i].y;
            As[buf][k + 2][r] = a_reg[i].z;
            As[buf][k + 3][r] = a_reg[i].w;
        }
#pragma unroll
        for (int i = 0; i < T::kBIters; ++i) {
            const int idx = tid + i * T::kThreads;
            const int krow = idx / (BN / 4);
            const int n4 = idx % (BN / 4);
            if (krow < BK) *reinterpret_cast<float4*>(&Bs[buf][krow][n4 * 4]) = b_reg[i];
        }
    };

    dr_f32x16 acc[T::kTM][T::kTN];
#pragma unroll
    for (int i = 0; i < T::kTM; ++i)
#pragma unroll
        for (int j = 0; j < T::kTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int lk = lane >> 5;          // which k of the pair this lane feeds
    const int li = lane & 31;
    for (int t = 0; t < T_total; ++t) {
        const int buf = t & 1;
        if (ABL != 1 && t + 1 < T_total) load_tile(t + 1);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[T::kTM], b[T::kTN];
#pragma unroll
            for (int i = 0; i < T::kTM; ++i) a[i] = As[buf][2 * kk + lk][wm * T::kWTM + i * 32 + li];
#pragma unroll
            for (int j = 0; j < T::kTN; ++j) b[j] = Bs[buf][2 * kk + lk][wn * T::kWTN + j * 32 + li];
#pragma unroll
            for (int i = 0; i < T::kTM; ++i)
#pragma unroll
                for (int j = 0; j < T::kTN; ++j) {
                    if (ABL == 2) acc[i][j][0] = fmaf(a[i], b[j], acc[i][j][0]);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
                }
        }
        if (ABL != 1 && t + 1 < T_total) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ----------------------------------------------------------------------------
    // D layout (32x32): col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < T::kTN; ++j) {
        const int n = n0 + wn * T::kWTN + j * 32 + li;
        const bool n_ok = n < p.Cout;
        const float sc = (n_ok && p.scale) ? p.scale[n] : 1.f;
        const float sh = (n_ok && p.shift) ? p.shift[n] : 0.f;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int i = 0; i < T::kTM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * T::kWTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
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
            if (lk == 0 && n_ok) {
                atomicAdd(&p.stat_sum[n], s1);
                atomicAdd(&p.stat_sq[n], s2);
            }
        }
    }
}

// Host-side launcher: picks the tile shape from (M, Cout).
int launch_conv_igemm(const ConvParams& p, hipStream_t stream);

}  // namespace dr
