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
#include <type_traits>

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
    const float* zeros;                              // >= 16 B of zeros in HBM: target of predicated-off loads
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
            int t = 0;                                     // no div/mod by the runtime ksize in here
            for (int dy = -pad; dy <= pad; ++dy)
                for (int dx = -pad; dx <= pad; ++dx, ++t) {
                    const int yy = y + dy, xx = x + dx;
                    if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) mask |= 1u << t;
                }
        }
        a_taps[i] = mask;
        a_off[i] = ok ? (unsigned)((long)m * p.x_cs + p.x_coff + a_k4[i] * 4) : 0u;
    }
    // weight tile: thread -> (k row, 4 columns); at most two float4 per thread (BN = 128).  Scalars, not arrays:
    // hipcc kept the two-element arrays in scratch once the K loop was unrolled by two.
    static_assert(T::kBIters <= 2, "B loader handles at most two float4 per thread");
    const int b_krow0 = tid / (BN / 4), b_n40 = tid % (BN / 4);
    const int b_krow1 = (tid + T::kThreads) / (BN / 4), b_n41 = (tid + T::kThreads) % (BN / 4);
    const bool b_ok0 = b_krow0 < BK && n0 + b_n40 * 4 < p.Np;
    const bool b_ok1 = T::kBIters > 1 && b_krow1 < BK && n0 + b_n41 * 4 < p.Np;
    const unsigned b_off0 = (unsigned)(b_krow0 * p.Np + n0 + b_n40 * 4);
    const unsigned b_off1 = (unsigned)(b_krow1 * p.Np + n0 + b_n41 * 4);

    float4 a_reg[T::kAIters];
    float4 b_reg0, b_reg1;

    // Refill = UNCONDITIONAL loads: a predicated-off lane reads 16 B of zeros from p.zeros (pointer select,
    // no branch).  With "v = 0; if (ok) v = load" hipcc copies the loaded value at the join and parks an
    // s_waitcnt vmcnt(0) right behind every load, stalling the wave in front of the MFMAs the prefetch was meant
    // to overlap.  The (tap, channel-chunk) cursor of the NEXT tile advances incrementally (no div/mod per tile).
    int ld_kc = 0, ld_dy = -pad, ld_dx = -pad, ld_tap = 0;
    const float* ld_x = p.x + (long)(ld_dy * p.W + ld_dx) * p.x_cs;      // wave-uniform cursors
    const float* ld_w = p.w;
    const bool ragged = (p.Cin & 3) != 0;                                  // uniform: Cin % 4 != 0
    int a_nv[T::kAIters];                                                  // only meaningful on ragged tiles
    auto load_tile = [&]() __attribute__((always_inline)) {
        const bool tail = ld_kc + BK > p.Cin;                              // uniform: this chunk crosses Cin
#pragma unroll
        for (int i = 0; i < T::kAIters; ++i) {
            bool ok = (a_taps[i] >> ld_tap) & 1u;
            int nv = 4;
            if (tail) {
                const int left = p.Cin - (ld_kc + a_k4[i] * 4);
                nv = left < 0 ? 0 : (left > 4 ? 4 : left);
                ok = ok && nv > 0;
            }
            const float* src = ok ? ld_x + ld_kc + a_off[i] : p.zeros;
            a_reg[i] = *reinterpret_cast<const float4*>(src);
            a_nv[i] = ok ? nv : 4;                                          // zeros need no masking
        }
        b_reg0 = *reinterpret_cast<const float4*>(b_ok0 ? ld_w + b_off0 : p.zeros);
        if constexpr (T::kBIters > 1) b_reg1 = *reinterpret_cast<const float4*>(b_ok1 ? ld_w + b_off1 : p.zeros);
        // advance the cursor
        ld_kc += BK;
        ld_w += (long)BK * p.Np;
        if (ld_kc >= p.Kp) {
            ld_kc = 0;
            ++ld_tap;
            if (++ld_dx > pad) { ld_dx = -pad; ++ld_dy; }
            ld_x = p.x + (long)(ld_dy * p.W + ld_dx) * p.x_cs;
        }
    };
    auto store_tile = [&](const int buf, bool was_tail) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < T::kAIters; ++i) {
            const int k = a_k4[i] * 4, r = a_row[i];
            float4 v = a_reg[i];
            if (ragged && was_tail) {                                       // uniform branch, rare layers only
                const int nv = a_nv[i];
                v.y = nv > 1 ? v.y : 0.f;
                v.z = nv > 2 ? v.z : 0.f;
                v.w = nv > 3 ? v.w : 0.f;
            }
            As[buf][k + 0][r] = v.x;
            As[buf][k + 1][r] = v.y;
            As[buf][k + 2][r] = v.z;
            As[buf][k + 3][r] = v.w;
        }
        if (b_krow0 < BK) *reinterpret_cast<float4*>(&Bs[buf][b_krow0][b_n40 * 4]) = b_reg0;
        if constexpr (T::kBIters > 1) {
            if (b_krow1 < BK) *reinterpret_cast<float4*>(&Bs[buf][b_krow1][b_n41 * 4]) = b_reg1;
        }
    };

    dr_f32x16 acc[T::kTM][T::kTN];
#pragma unroll
    for (int i = 0; i < T::kTM; ++i)
#pragma unroll
        for (int j = 0; j < T::kTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    bool tail0 = BK > p.Cin;
    load_tile();
    store_tile(0, tail0);
    __syncthreads();

    const int lk = lane >> 5;          // which k of the pair this lane feeds
    const int li = lane & 31;
    // The K loop is unrolled by two so that the LDS buffer index is a compile-time constant: every ds_read /
    // ds_write address is then "base + immediate" instead of a per-access VALU add.
    for (int t = 0; t < T_total; t += 2) {
#pragma unroll
        for (int buf = 0; buf < 2; ++buf) {
            if (t + buf < T_total) {
                const bool more = ABL != 1 && t + buf + 1 < T_total;
                const bool was_tail = ld_kc + BK > p.Cin;                   // of the tile being fetched now
                if (more) load_tile();
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
                if (more) store_tile(buf ^ 1, was_tail);
                __syncthreads();
            }
        }
    }

    // ---- epilogue ----------------------------------------------------------------------------
    // D layout (32x32): col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    // Every global read of the epilogue (output row mask, dropout keep bytes, residual) is issued as a batch of
    // independent, unconditional loads BEFORE the first use (a masked-off lane reads element 0 and ignores it):
    // written as "if (ok) v += res[...]" per element, hipcc serialises them behind s_waitcnt vmcnt(0) -- 16
    // dependent HBM round trips per 32x32 tile, which was most of the run time of the short-K 1x1 layers.
    // Indices are 32-bit element offsets from the (uniform) tensor base, so each access is "sgpr base + vgpr
    // offset" and costs one address register; launch_conv_igemm rejects tensors of 2^32 elements or more.
    double s1[T::kTN], s2[T::kTN];
#pragma unroll
    for (int j = 0; j < T::kTN; ++j) s1[j] = s2[j] = 0.0;
    const bool dropping = p.drop || p.drop_rng;
#pragma unroll
    for (int i = 0; i < T::kTM; ++i) {
        const int mb = m0 + wm * T::kWTM + i * 32 + 4 * lk;               // row of r = 0
        unsigned rows = 0;                                                  // bit r: this lane writes row r
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (mb + (r & 3) + 8 * (r >> 2) < M) rows |= 1u << r;
        if (p.out_rowmask) {
            float om[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) om[r] = p.out_rowmask[((rows >> r) & 1u) ? (unsigned)(mb + (r & 3) + 8 * (r >> 2)) : 0u];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (om[r] < p.out_mask_thresh) rows &= ~(1u << r);
        }
#pragma unroll
        for (int j = 0; j < T::kTN; ++j) {
            const int n = n0 + wn * T::kWTN + j * 32 + li;
            const bool n_ok = n < p.Cout;
            const float sc = (n_ok && p.scale) ? p.scale[n] : 1.f;
            const float sh = (n_ok && p.shift) ? p.shift[n] : 0.f;
            const unsigned rows_j = n_ok ? rows : 0u;
#pragma unroll
            for (int half = 0; half < 2; ++half) {                         // 8 rows at a time: bounds the VGPR peak
                float rv[8];
                unsigned keep = 0xFFu;
                if (p.res) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int r = half * 8 + q;
                        const unsigned m = (unsigned)(mb + (r & 3) + 8 * (r >> 2));
                        rv[q] = p.res[((rows_j >> r) & 1u) ? m * (unsigned)p.res_cs + (unsigned)(p.res_coff + n) : 0u];
                    }
                }
                if (p.drop) {
                    unsigned char dv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int r = half * 8 + q;
                        const unsigned m = (unsigned)(mb + (r & 3) + 8 * (r >> 2));
                        dv[q] = p.drop[((rows_j >> r) & 1u) ? m * (unsigned)p.Cout + (unsigned)n : 0u];
                    }
                    keep = 0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) keep |= (dv[q] ? 1u : 0u) << q;
                } else if (p.drop_rng) {
                    keep = 0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int r = half * 8 + q;
                        const unsigned long long m = (unsigned long long)(mb + (r & 3) + 8 * (r >> 2));
                        keep |= (dropout_keep(p.drop_seed, m * p.Cout + n) ? 1u : 0u) << q;
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = half * 8 + q;
                    if (ABL == 3 && acc[i][j][r] != 12345.678f) continue;
                    if ((rows_j >> r) & 1u) {
                        const unsigned m = (unsigned)(mb + (r & 3) + 8 * (r >> 2));
                        const float raw = acc[i][j][r];
                        s1[j] += (double)raw;
                        s2[j] += (double)raw * (double)raw;
                        float v = raw * sc + sh;
                        if (p.relu) v = fmaxf(v, 0.f);
                        if (dropping) v = ((keep >> q) & 1u) ? v * 2.f : 0.f;
                        if (p.res) v += rv[q];
                        p.y[m * (unsigned)p.y_cs + (unsigned)(p.y_coff + n)] = v;
                    }
                }
            }
        }
    }
    if (p.stat_sum) {
#pragma unroll
        for (int j = 0; j < T::kTN; ++j) {
            const int n = n0 + wn * T::kWTN + j * 32 + li;
            double a = s1[j], b = s2[j];
            a += __shfl_xor(a, 32);
            b += __shfl_xor(b, 32);
            if (lk == 0 && n < p.Cout) {
                atomicAdd(&p.stat_sum[n], a);
                atomicAdd(&p.stat_sq[n], b);
            }
        }
    }
}

// Host-side launcher: picks the tile shape from (M, Cout).
int launch_conv_igemm(const ConvParams& p, hipStream_t stream);

}  // namespace dr
