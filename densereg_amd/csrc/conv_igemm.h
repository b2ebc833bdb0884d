// conv_igemm.h -- implicit-GEMM stride-1 SAME convolution (1x1 / 3x3) on the fp32 matrix cores.
//
// Replaces tf.nn.conv2d + BatchReNorm/bias + ReLU (+ residual add, + dropout) as composed by
// network/slim/ops.py:219-299 and network/um_v1.py:18-48 (reference, NHWC x HWIO, fp32).
//
//   GEMM view:  M = B*H*W output pixels,  N = Cout,  K = taps*Cin
//   A[m][k]   = x[b, y+dy, x+dx, c]   gathered on the fly (zero outside the image: TF 'SAME')
//   B[k][n]   = packed weights [Kp/16][tap][Np][16]: the K-tile order of the main loop, so one weight tile
//               (BN rows x 16 k) is one contiguous block (HWIO transposed, Cin->Kp, Cout->Np zero padded)
//   D         = v_mfma_f32_32x32x2_f32 chains: exact fp32 (one rounding per product, fixed k order),
//               157 TFLOP/s peak on gfx950 -- the roofline this kernel is priced against.
//
// Block = 256 threads = 4 waves; wave tile = (BM/WM) x (BN/WN) made of 32x32 MFMA tiles.
// LDS holds As[BM][BK+4] and Bs[BN][BK+4], both k-CONTIGUOUS -- the layout the operands have in HBM, so a
// refill is one 16-byte global load and one ds_write_b128 per thread and operand.  The MFMA sums over two k
// slots (lane>>5); which two k values a step pairs up is free, so lane (i, lk) fetches k = 8g + 4*lk + {0..3}
// with ONE ds_read_b128 and feeds them to four consecutive MFMA steps: step s multiplies k = 8g+s and
// 8g+4+s.  That is a quarter of the LDS instructions of a dword-per-step fragment read.  Rows are unpadded
// (BK floats); the 16-byte slot of (row, k4) is XOR-swizzled to k4 ^ ((row >> 2) & 3), which makes both the
// ds_read_b128 lane groups (16 rows, one k4) and the ds_write_b128 groups (2 rows x 4 k4) bank-conflict free
// (SQ_LDS_BANK_CONFLICT was a third of the LDS cycles with a padded, unswizzled layout).
// Both tiles are double buffered with register prefetch of the next K-tile, one barrier per K-tile.
//
// Epilogue (per lane = one output channel, 16 rows):  v = acc*scale[n] + shift[n]; relu;
// dropout keep mask (x2); + residual;  optional per-channel sum / sum-of-squares of the RAW
// accumulator for train-mode BatchReNorm (tf.nn.moments, ops.py:132): one fp64 partial row per workgroup,
// folded in a fixed order by bn_fwd_finalize_kernel (no floating-point atomics: same-address fp64 atomics
// serialise at ~90 ns each on this part, and a fixed order makes the training step reproducible).
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
    double* stat_part;                               // nullable: [2][Cout][gx] per-workgroup sum / sum-of-squares of acc
    const float* zeros;                              // >= 16 B of zeros in HBM: target of predicated-off loads
    // dgrad feeding a BatchReNorm layer whose output has no other consumer (bst_raw != null; needs stat_part, no
    // residual/dropout): the values written ARE that layer's complete dOut, so its backward reduction happens here --
    // stat_part rows hold sum(g) and sum(g*yhat), g = dOut * [raw*scale+shift > 0], yhat = (raw-mean)*inv_std, instead
    // of the forward moments.  bst_raw is the layer's raw conv output (row stride bst_cs), bst_bnc = [mean | inv_std].
    const float* bst_raw; int bst_cs; const float* bst_scale; const float* bst_shift; const float* bst_bnc; int bst_relu;
    // bst_act != 0 -- the same idea for a BIAS conv with ReLU / dropout whose output has no other consumer (um_full 1/2):
    // bst_raw is that layer's OUTPUT (post ReLU and dropout), bst_scale / bst_shift / bst_bnc are null; the epilogue writes
    // g = dOut * bst_factor * [out > 0] (the gradient wrt the conv's pre-activation: its act_bwd pass disappears) and the
    // stat_part rows hold sum(g) = the bias gradient (its column-sum pass disappears too).
    int bst_act; float bst_factor;
    // micro-batch groups (train_kernels.h, BnTrainParams): the output rows are consecutive micro-batches of grp_rows rows (a
    // multiple of every tile's rows, so a workgroup lies in one group); group g's bst_scale / bst_shift are grp_fold floats
    // behind group g-1's, its bst_bnc 2 * grp_fold floats.  0: one batch.
    int grp_rows, grp_fold;
    int Ng;                                // > 0: compute only the first Ng (multiple of 32, <= Np) output columns -- Np
                                           // stays the row stride of the packed weights (input gradients of a concat
                                           // buffer whose last channels have no consumer)
    int gx, gy;                            // the launch grid (set by the launcher): read with the other arguments instead of from the
                                           // implicit arguments, whose loads sat serialised behind branches in every workgroup's prologue
    int nfast;                             // workgroup -> tile mapping: the N blocks of a row block are consecutive in dispatch order
    int stagger;                           // conv_x3.h: the second workgroup of every CU in the FIRST round of the grid starts this many s_sleep(127) late (0: off)
    int bf16;                              // w holds bf16 [Kp/32][tap][Np][32] (Kp % 32 == 0): launch the BF kernels
    int x_bf16;                            // BF kernels only: x holds bf16 elements (x_cs / x_coff in elements, channel groups of 4
                                           // zero-padded): a 16-byte slot is loaded as it is, no conversion while staging
    // BF kernels only (the epilogue's EP_IO16): the raw output of a BatchReNorm conv stored as bf16 elements (y_cs / y_coff in
    // elements; plain output only: no scale / shift / relu / res / dropout / bst) -- the statistics rows are sums over the ROUNDED
    // values, so the layer normalises exactly what it stored; and the producer's raw output read back that way by bst mode.
    // y_bf16 together with bst_raw_bf16: the input gradient this launch is the only writer of (no res), stored as bf16 elements.
    int y_bf16, bst_raw_bf16;
    // conv_x3.h: the same weights as three bf16 planes [Kp/16][tap][3][Np][16] (w = w0 + w1 + w2 to fp32 accuracy), or null.  With
    // it the launcher may run the layer on the bf16 matrix cores with fp32-accurate products (launch_conv_igemm: conv_use_x3).
    const void* w3;
    // conv_p3.h: the input STORED as its three bf16 planes ("P3": [M][xp3_cp/16][3][16] bf16, xp3_cp = Cin rounded up to 16, pad
    // channels zero; the whole tensor: no channel offset), or null.  With it (and w3) the launcher runs conv_p3_kernel where its tile
    // applies: the same products as conv_x3_kernel without the operand split in the K loop.  x stays valid or null (p.xp3 wins).
    const void* xp3; int xp3_cp;
};

// stateless keep bit for dropout(0.5): splitmix64 finaliser of (seed, element index)
__device__ __forceinline__ bool dropout_keep(unsigned long long seed, unsigned long long idx) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (z >> 17) & 1ull;
}

template <int BM, int BN, int WM, int WN, int BK_ = 16, int WK_ = 1, int MF_ = 32>
struct ConvTile {
    static constexpr int kBK = BK_;
    static constexpr int kThreads = 256;
    static constexpr int kSK = BK_;               // LDS row stride (floats) of both operand tiles: unpadded, swizzled
    static_assert(BK_ == 16 || BK_ == 64, "slot swizzles exist for 4 and 16 slots per row");
    // float offset to XOR into a row's k index: 4 slots -> slot ^ ((row>>2)&3); 16 slots (a 256-byte row = one full
    // bank row, so the 16 rows of a ds_read_b128 lane group must land on 16 different slots) -> slot ^ (row&15)
    __host__ __device__ static constexpr int swz(int row) { return BK_ == 16 ? (row & 12) : ((row & 15) << 2); }
    static constexpr int kWTM = BM / WM;          // wave tile rows
    static constexpr int kWTN = BN / WN;
    static constexpr int kTM = kWTM / MF_;        // MFMA output tiles of the wave tile (MF_ = 32: 32x32x2, 16: 16x16x4)
    static constexpr int kTN = kWTN / MF_;
    static constexpr int kAIters = (BM * (kBK / 4)) / kThreads;
    static constexpr int kBIters = (BN * (kBK / 4) + kThreads - 1) / kThreads;
    static_assert(WM * WN * WK_ == 4, "4 waves per block");
    static_assert(kWTM % MF_ == 0 && kWTN % MF_ == 0, "wave tile must be made of whole MFMA tiles");
    static_assert((BM * (kBK / 4)) % kThreads == 0, "A loader mapping");
};

// ABL (profiling ablations, product code uses 0): 1 = no global->LDS refills after the first K-tile,
// 2 = MFMA replaced by one VALU fma per fragment pair, 3 = no epilogue stores, 4 = refill loads issued and
// waited for but not written to LDS, 5 = LDS refill writes (of stale registers) without the global loads.
// Resident waves per SIMD the register allocator must leave room for.  M = 40960 rows (B=40 at 32x32) gives
// 640 row blocks of 64: with Np = 256 / 512 that is exactly 5 / 10 workgroups per CU, so five resident
// workgroups finish in whole rounds, while four leave a last round with one lonely workgroup per CU whose
// load latency nothing hides (measured: a quarter of the kernel time).
//
// BK_ = 64 ("fat K-tile") is the variant for grids that cannot fill the chip (everything below 32x32): such a launch
// is a serial chain of K-tiles at ~0.6 us each (load -> LDS -> barrier -> MFMA, one wave per SIMD, nothing to hide
// behind), so it moves four 16-channel chunks per round trip -- a 3x3 64->64 layer is 9 iterations instead of 36.
template <int BM, int BN, int BK_>
constexpr int conv_min_waves() { return BK_ == 64 ? 2 : (BM * BN >= 128 * 128 ? 3 : (BM == 128 && BN == 64 ? 4 : 5)); }

// GL = 1: the refill is an LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write pass, the copy lands
// while the MFMAs of the current tile run and is retired by the barrier's vmcnt(0).  The LDS image must be lane-linear,
// so the slot swizzle moves to the SOURCE address (lane (row, s) fetches chunk s ^ f(row)); needs Cin % 4 == 0 (a
// 16-byte chunk is copied whole or replaced by the zero page).  Measured +6..8 % on every shape in isolation (3x3 256->256:
// 478 -> 448 us); the launcher's default since round 2 (DR_CONV_GLDS=0 selects the register-staged refill).
//
// BF = 1: bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate) on fp32 tensors.  The LDS image keeps its
// geometry -- 64-byte rows, four swizzled 16-byte slots -- but a slot now holds 8 bf16 channels, so a K-tile is 32
// input channels and one ds_read_b128 per operand feeds an MFMA with 16 k (lane half lk supplies k 8*lk..8*lk+7 of
// both operands, the same freedom to permute k the fp32 path uses).  Activations are converted while they are staged
// (v_cvt_pk_bf16_f32, round to nearest even); weights arrive packed as bf16 [Kp/32][tap][Np][32] (pack_all_kernel),
// byte for byte the fp32 tile layout, so the weight loader is unchanged.  Epilogue, masks and views are the fp32 ones.
// WK = 2 (narrow outputs: N = 65..96 and 129..160, tiles 64x96 and 64x160): the four waves are 2 (rows) x 2 (K halves).
// A wave owns 32 rows x ALL BN columns (3 or 5 accumulator tiles) and the k-slots 8*wk .. 8*wk+7 of every K-tile; the two
// halves are summed through LDS after the K loop, wave wk = 0 runs the epilogue.  These layers ran on the 128x32 tile, whose
// 32-column blocks re-stage the A tile three or five times and leave each wave ONE accumulator chain (0.27 of the MFMA
// roofline); here A is staged once, 640 workgroups of M = 40960 fall 2.5 per CU, and every fragment pair feeds 3 or 5 MFMAs.
// XB bit 1 (BF kernels; XB = 2 / 3): the kernel carries the two extra copies of the epilogue for bf16-stored raw outputs and
// gradients (ConvParams::y_bf16 / bst_raw_bf16; conv_epilogue.inc) -- the launcher picks these instantiations for the training
// launches that need them, everything else (the eval path, bias convs, plain input gradients) runs the one-copy kernels: with
// the copies in every bf16 kernel the eval-mode forward lost 2-4 % (visit 16: one engine 15 540 -> 14 870 crops/s).
// XB = 1 (BF kernels): the A operand is stored as bf16 (ConvParams::x_bf16) -- a compile-time variant, because a run-time
// branch around the staging loads keeps hipcc from issuing them as one batch.
// MF = 16 (narrow outputs in fp32: N = 65..80, 129..160; tiles 64x80, 64x144, 64x160): v_mfma_f32_16x16x4_f32 instead of 32x32x2 --
// the same arithmetic per product and the same rate (64 flops per clock and SIMD), but output tiles of 16 columns, so a 78-
// channel layer computes 80 columns instead of 96 and a 131-channel one 144 instead of 160.  The four waves are 4 row groups of
// 16; each owns ALL BN columns (5 / 9 / 10 independent accumulator tiles): the A tile is staged once per workgroup where the
// 128x32 tile re-staged it for every 32-column block and left each wave one accumulator chain.  Lane (r = lane & 15,
// g = lane >> 4) feeds row / column r and, in MFMA step s, k = 4g + s: like the 32x32 path it fetches its four k with ONE
// ds_read_b128 per operand tile and K-tile (which k a step pairs up is free as long as A and B agree).
template <int BM, int BN, int WM, int WN, int ABL = 0, int BK_ = 16, int GL = 0, int BF = 0, int WK = 1, int XB = 0, int MF = 32>
__global__ __launch_bounds__(256, (MF == 16 ? 5 : WK > 1 ? (BN > 96 ? 3 : 4) : conv_min_waves<BM, BN, BK_>())) void conv_igemm_kernel(const ConvParams p) {
    static_assert(MF == 32 || (MF == 16 && BM == 64 && WM == 4 && WN == 1 && BK_ == 16 && GL == 0 && BF == 0 && WK == 1 && XB == 0 && ABL == 0),
                  "16x16x4 variant: 64 rows x BN columns, fp32, register-staged refill");
    static_assert(XB == 0 || BF == 1, "bf16 storage of the A operand exists for the bf16 matrix-core kernels");
    static_assert(GL == 0 || (BK_ == 16 && BN >= 64 && BN % 64 == 0), "the LDS-DMA refill needs every wave's 64 lanes inside both tiles");
    static_assert(BF == 0 || (GL == 0 && BK_ == 16 && ABL == 0), "the bf16 variant exists for the register-staged 64-byte-row tile");
    static_assert(WK == 1 || (WK == 2 && BK_ == 16 && ABL == 0 && GL == 0), "K-split: two halves of a 16-k tile");
    using T = ConvTile<BM, BN, WM, WN, BK_, WK, MF>;
    constexpr int BK = T::kBK;
    constexpr int SK = T::kSK;
    constexpr int CK = BF ? 32 : BK;      // input channels per K-tile
    constexpr int CS = BF ? 8 : 4;        // input channels per 16-byte LDS slot
    // The two pipeline stages are SEPARATE LDS objects (selected by a compile-time stage index), not one [2][..] array:
    // hipcc's wait-count insertion tells LDS accesses apart by the alias scope the LDS lowering gives each object, and
    // with a single array it put "s_waitcnt vmcnt(0)" between a tile's LDS-DMA (GL) and the fragment reads of the OTHER
    // stage -- the asynchronous copy was waited for before the first MFMA it was meant to run under.
    __shared__ __attribute__((aligned(16))) float As0[BM][SK];
    __shared__ __attribute__((aligned(16))) float As1[BM][SK];
    __shared__ __attribute__((aligned(16))) float Bs0[BN][SK];
    __shared__ __attribute__((aligned(16))) float Bs1[BN][SK];
#define DR_AS(stage) ((stage) ? As1 : As0)
#define DR_BS(stage) ((stage) ? Bs1 : Bs0)

    DR_PIN_ARGS(p.x, p.x_cs, p.x_coff, p.Cin, p.B, p.H, p.W, p.ksize, p.w, p.Kp, p.Np, p.rowmask, p.zeros, p.nfast, p.gx, p.gy, p.Ng);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wk = WK > 1 ? wave / (WM * WN) : 0;          // K half of this wave (WK = 2)
    const int wm = (wave % (WM * WN)) / WN;
    const int wn = wave % WN;
    const int HW = p.H * p.W;
    const int M = p.B * HW;
    // XCD-aware row-block mapping: workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own
    // 4 MB L2.  A 3x3 tap reads the image rows above and below a row block, i.e. its neighbours' pixels: with the
    // identity mapping neighbours sit on different XCDs and every L2 fetches every halo from HBM.  Remapped, XCD x
    // owns the contiguous range [x*nx/8, (x+1)*nx/8) of row blocks (and all their N blocks: gridDim.x % 8 == 0).
    // N blocks of one row block are dealt out back to back (p.nfast): they run at the same time on the same XCD, so the row
    // block's input pixels are fetched from HBM once and re-read from L2 by the other N blocks.  With the dispatch order of a
    // 2-D grid (x fastest) the N blocks of a row block start gridDim.x workgroups apart and each of them reads the rows from HBM.
    const int gx = p.gx, gy = p.gy;
    int mblk = blockIdx.x, nblk = blockIdx.y;
    if (p.nfast && gy > 1) {
        const int L = blockIdx.y * gx + blockIdx.x, nN = gy;
        if ((gx & 7) == 0) { const int s = L >> 3; mblk = (L & 7) * (gx >> 3) + s / nN; nblk = s % nN; }
        else { mblk = L / nN; nblk = L % nN; }
    } else if ((gx & 7) == 0) {
        mblk = (blockIdx.x & 7) * (gx >> 3) + (blockIdx.x >> 3);
    }
    const int m0 = mblk * BM;
    const int n0 = nblk * BN;
    const int taps = p.ksize * p.ksize;
    const int KT = (p.Kp + CK - 1) / CK;
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
    // image sides are powers of two on this network: shifts instead of three integer divisions per row; the tap
    // mask is assembled branch-free from "has a row above / below, a column left / right" (this prologue runs
    // in every workgroup and was ~2 us of dependent integer code per owned row)
    const bool pow2 = (p.W & (p.W - 1)) == 0 && (HW & (HW - 1)) == 0;
    const int w_shift = __builtin_ctz((unsigned)p.W);
#pragma unroll
    for (int i = 0; i < T::kAIters; ++i) {
        const int idx = tid + i * T::kThreads;
        const int row = idx / (BK / 4);
        a_row[i] = row;
        a_k4[i] = idx % (BK / 4);
        if (GL) a_k4[i] ^= (row >> 2) & 3;                             // LDS-DMA: the slot swizzle lives on the source side
        const int m = m0 + row;
        bool ok = m < M;
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
        a_off[i] = ok ? (unsigned)((long)m * p.x_cs + p.x_coff + a_k4[i] * CS) : 0u;
    }
    // weight tile.  BK = 16: thread -> (output channel row, 4 consecutive k), at most two float4 per thread (BN =
    // 128); scalars, not arrays: hipcc kept two-element arrays in scratch once the K loop was unrolled by two.
    // BK = 64: the tile is four 16-channel chunks, each a contiguous [BN][16] block of the packed weights; iteration
    // i of a thread IS chunk i (256 threads = 64 rows x 4 k4), so one row / k4 / offset serves all four.
    static_assert(BK == 64 ? (BN == 64) : (T::kBIters <= 3), "B loader mapping");
    constexpr int BKC = 16;                                                 // packing granularity of the weights
    const int b_row0 = tid / (BKC / 4), b_row1 = (tid + T::kThreads) / (BKC / 4), b_k4 = tid % (BKC / 4);
    const bool b_ok0 = b_row0 < BN && n0 + b_row0 < p.Np;
    const bool b_ok1 = BK == 16 && T::kBIters > 1 && b_row1 < BN && n0 + b_row1 < p.Np;
    const int b_row2 = (tid + 2 * T::kThreads) / (BKC / 4);                 // third float4 (BN = 160 only)
    const bool b_ok2 = BK == 16 && T::kBIters > 2 && b_row2 < BN && n0 + b_row2 < p.Np;
    const unsigned b_off2 = (unsigned)((n0 + b_row2) * BKC + b_k4 * 4);
    const unsigned b_off0 = (unsigned)((n0 + b_row0) * BKC + b_k4 * 4);     // = n0*16 + 4*tid: fully coalesced
    const unsigned b_off1 = (unsigned)((n0 + b_row1) * BKC + b_k4 * 4);
    const long w_chunk = (long)taps * p.Np * BKC;                           // floats between consecutive chunks of a tap
    const int n_chunks = p.Kp / BKC;

    float4 a_reg[T::kAIters];
    float4 a_hi[BF ? T::kAIters : 1];                                      // BF: channels 4..7 of the slot
    float4 b_reg0, b_reg1, b_reg2;
    float4 b_fat0, b_fat1, b_fat2, b_fat3;                                 // BK = 64: one float4 per chunk (named: a
                                                                           // float4[4] here was promoted to LDS by hipcc)

    // Refill = UNCONDITIONAL loads: a predicated-off lane reads 16 B of zeros from p.zeros (pointer select,
    // no branch).  With "v = 0; if (ok) v = load" hipcc copies the loaded value at the join and parks an
    // s_waitcnt vmcnt(0) right behind every load, stalling the wave in front of the MFMAs the prefetch was meant
    // to overlap.  The (channel-chunk, tap) cursor of the NEXT tile advances incrementally (no div/mod per tile).
    int ld_kc = 0, ld_dy = -pad, ld_dx = -pad, ld_tap = 0;
    const float* ld_x = p.x + (long)(ld_dy * p.W + ld_dx) * p.x_cs;      // wave-uniform cursors
    long ld_xo = (long)(ld_dy * p.W + ld_dx) * p.x_cs;                    // the same shift as an element offset (bf16 sources)
    constexpr bool xb16 = (XB & 1) != 0;
    const float* ld_w = p.w;
    const bool ragged = (p.Cin & 3) != 0;                                  // uniform: Cin % 4 != 0
    int a_nv[T::kAIters];                                                  // only meaningful on ragged tiles
    auto load_tile = [&](const int dst) __attribute__((always_inline)) {
        if constexpr (GL) {
#pragma unroll
            for (int i = 0; i < T::kAIters; ++i) {
                const bool ok = ((a_taps[i] >> ld_tap) & 1u) && ld_kc + a_k4[i] * 4 < p.Cin;   // Cin % 4 == 0: whole chunks
                dr_glds16(ok ? ld_x + ld_kc + a_off[i] : p.zeros, &DR_AS(dst)[0][0] + (wave * 64 + i * T::kThreads) * 4);
            }
#pragma unroll
            for (int i = 0; i < T::kBIters; ++i) {
                const int idx = tid + i * T::kThreads;
                const int brow = idx >> 2, bs = (idx & 3) ^ ((brow >> 2) & 3);
                const bool ok = brow < BN && n0 + brow < p.Np;
                dr_glds16(ok ? ld_w + (unsigned)((n0 + brow) * BKC + bs * 4) : p.zeros, &DR_BS(dst)[0][0] + (wave * 64 + i * T::kThreads) * 4);
            }
        }
        const bool tail = !GL && ld_kc + CK > p.Cin;                       // uniform: this chunk crosses Cin
        if constexpr (!GL) {
#pragma unroll
        for (int i = 0; i < T::kAIters; ++i) {
            bool ok = (a_taps[i] >> ld_tap) & 1u;
            int nv = CS;
            if (tail) {
                const int left = p.Cin - (ld_kc + a_k4[i] * CS);
                nv = left < 0 ? 0 : (left > CS ? CS : left);
                ok = ok && nv > 0;
            }
            if constexpr (xb16) {                                           // 8 bf16 channels = one slot, as stored
                const __bf16* sb = reinterpret_cast<const __bf16*>(p.x) + (ld_xo + ld_kc + (long)a_off[i]);
                a_reg[i] = dr_load16_a4(ok ? reinterpret_cast<const void*>(sb) : reinterpret_cast<const void*>(p.zeros));
                a_nv[i] = ok ? nv : CS;
                continue;
            }
            const float* src = ok ? ld_x + ld_kc + a_off[i] : p.zeros;
            a_reg[i] = *reinterpret_cast<const float4*>(src);
            if constexpr (BF) a_hi[i] = *reinterpret_cast<const float4*>(ok && nv > 4 ? src + 4 : p.zeros);
            a_nv[i] = ok ? nv : CS;                                         // zeros need no masking
        }
        if constexpr (BK == 16) {
            b_reg0 = *reinterpret_cast<const float4*>(b_ok0 ? ld_w + b_off0 : p.zeros);
            if constexpr (T::kBIters > 1) b_reg1 = *reinterpret_cast<const float4*>(b_ok1 ? ld_w + b_off1 : p.zeros);
            if constexpr (T::kBIters > 2) b_reg2 = *reinterpret_cast<const float4*>(b_ok2 ? ld_w + b_off2 : p.zeros);
        } else {
            const int left = n_chunks - ld_kc / BKC;                        // the last chunk group may be short
            const float* w0 = ld_w + b_off0;
            b_fat0 = *reinterpret_cast<const float4*>(b_ok0 && left > 0 ? w0 : p.zeros);
            b_fat1 = *reinterpret_cast<const float4*>(b_ok0 && left > 1 ? w0 + w_chunk : p.zeros);
            b_fat2 = *reinterpret_cast<const float4*>(b_ok0 && left > 2 ? w0 + 2 * w_chunk : p.zeros);
            b_fat3 = *reinterpret_cast<const float4*>(b_ok0 && left > 3 ? w0 + 3 * w_chunk : p.zeros);
        }
        }
        // advance the cursor: taps innermost.  The nine taps of one 16-channel chunk re-read the same 64-byte
        // pixel slices (shifted by a pixel), one K-tile apart, so they hit in L1/L2; with the channel sweep
        // innermost the re-read came 16 K-tiles later, after the slice had left this XCD's 4 MB L2.
        ++ld_tap;
        if (++ld_dx > pad) { ld_dx = -pad; ++ld_dy; }
        if (ld_tap == taps) {
            ld_tap = 0;
            ld_dy = ld_dx = -pad;
            ld_kc += CK;
        }
        ld_x = p.x + (long)(ld_dy * p.W + ld_dx) * p.x_cs;
        ld_xo = (long)(ld_dy * p.W + ld_dx) * p.x_cs;
        if constexpr (BK == 16) ld_w += p.Np * BKC;                        // packed in exactly this order
        else ld_w = p.w + ((long)(ld_kc / BKC) * taps + ld_tap) * p.Np * BKC;
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
            if constexpr (xb16) {
                // stored bf16: channel groups of four are zero-padded by their producer, so only a slot whose upper half lies
                // beyond the row's channels needs masking
                float4 w16 = a_reg[i];
                if (was_tail && a_nv[i] <= 4) { w16.z = 0.f; w16.w = 0.f; }
                *reinterpret_cast<float4*>(&DR_AS(buf)[r][k ^ T::swz(r)]) = w16;
                continue;
            }
            if constexpr (BF) {
                float4 u = a_hi[i];
                if (ragged && was_tail) {
                    const int nv = a_nv[i];
                    u.y = nv > 5 ? u.y : 0.f;
                    u.z = nv > 6 ? u.z : 0.f;
                    u.w = nv > 7 ? u.w : 0.f;
                }
                const dr_f32x8 f = {v.x, v.y, v.z, v.w, u.x, u.y, u.z, u.w};
                v = __builtin_bit_cast(float4, __builtin_convertvector(f, dr_bf16x8));
            }
            *reinterpret_cast<float4*>(&DR_AS(buf)[r][k ^ T::swz(r)]) = v;          // swizzled 16-byte slot
        }
        if constexpr (BK == 16) {
            if (b_row0 < BN) *reinterpret_cast<float4*>(&DR_BS(buf)[b_row0][(b_k4 * 4) ^ T::swz(b_row0)]) = b_reg0;
            if constexpr (T::kBIters > 1) {
                if (b_row1 < BN) *reinterpret_cast<float4*>(&DR_BS(buf)[b_row1][(b_k4 * 4) ^ T::swz(b_row1)]) = b_reg1;
            }
            if constexpr (T::kBIters > 2) {
                if (b_row2 < BN) *reinterpret_cast<float4*>(&DR_BS(buf)[b_row2][(b_k4 * 4) ^ T::swz(b_row2)]) = b_reg2;
            }
        } else {
            float* brow = &DR_BS(buf)[b_row0][0];
            const int sw = T::swz(b_row0), kq = b_k4 * 4;
            *reinterpret_cast<float4*>(brow + ((0 * BKC + kq) ^ sw)) = b_fat0;
            *reinterpret_cast<float4*>(brow + ((1 * BKC + kq) ^ sw)) = b_fat1;
            *reinterpret_cast<float4*>(brow + ((2 * BKC + kq) ^ sw)) = b_fat2;
            *reinterpret_cast<float4*>(brow + ((3 * BKC + kq) ^ sw)) = b_fat3;
        }
    };

    // A wave that owns a single 32x32 output tile would run ONE chain of dependent MFMAs (each waits for the
    // previous result: ~150 cycles instead of the 64-cycle issue rate, and on a grid that leaves one wave per SIMD
    // nothing fills the gap).  Such tiles accumulate alternate k-steps into KACC independent accumulators that are
    // summed, in a fixed order, after the K loop.
    constexpr int KACC = (T::kTM * T::kTN == 1) ? (BK == 64 ? 4 : 2) : 1;
    using AccT = typename std::conditional<MF == 16, dr_f32x4, dr_f32x16>::type;
    constexpr int NR = MF == 16 ? 4 : 16;              // accumulator registers per lane and MFMA tile
    AccT accp[KACC][T::kTM][T::kTN];
#pragma unroll
    for (int q = 0; q < KACC; ++q)
#pragma unroll
        for (int i = 0; i < T::kTM; ++i)
#pragma unroll
            for (int j = 0; j < T::kTN; ++j)
#pragma unroll
                for (int r = 0; r < NR; ++r) accp[q][i][j][r] = 0.f;

    bool tail0 = CK > p.Cin;
    load_tile(0);
    if constexpr (!GL) store_tile(0, tail0);
    __syncthreads();

    float abl_sink = 0.f;              // ABL 4 only
    const float abl_frag = (ABL == 7 || ABL >= 8) ? p.mask_thresh + (float)lane * 1e-3f : 0.f;   // run-time value: nothing folds away
    const int lk = lane >> 5;          // which k of the pair this lane feeds
    const int li = lane & 31;
    // One K-tile: prefetch tile t+1 into registers, MFMA over tile t from LDS buffer `buf`, park t+1 in the other
    // buffer, barrier.  `buf` is a compile-time constant in every call (the K loop below is unrolled by two), so
    // each ds_read / ds_write address is "base + immediate" instead of a per-access VALU add.
    auto k_tile = [&](const int buf, const bool more_) __attribute__((always_inline)) {
        const bool more = ABL != 1 && ABL < 6 && more_;                   // ABL 6..8: no refills, and (6) no barrier, (7) fragments read once, (8) both
        const bool was_tail = ld_kc + CK > p.Cin;                           // of the tile being fetched now
        if (more && ABL != 5) load_tile(buf ^ 1);
        if constexpr (MF == 16) {
            // lane (r16, g): row wm*16 + r16 of A, rows j*16 + r16 of B, the four k of slot g
            const int r16 = lane & 15, g16 = lane >> 4;
            const float4 a16 = *reinterpret_cast<const float4*>(&DR_AS(buf)[wm * 16 + r16][(g16 * 4) ^ T::swz(r16)]);
#pragma unroll
            for (int j = 0; j < T::kTN; ++j) {
                const float4 b16 = *reinterpret_cast<const float4*>(&DR_BS(buf)[j * 16 + r16][(g16 * 4) ^ T::swz(r16)]);
                accp[0][0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a16.x, b16.x, accp[0][0][j], 0, 0, 0);
                accp[0][0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a16.y, b16.y, accp[0][0][j], 0, 0, 0);
                accp[0][0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a16.z, b16.z, accp[0][0][j], 0, 0, 0);
                accp[0][0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a16.w, b16.w, accp[0][0][j], 0, 0, 0);
            }
            if (more) store_tile(buf ^ 1, was_tail);
            __syncthreads();
        } else {
        // WK = 2: this wave multiplies only the k-slot group g = wk of the tile (the other half belongs to its partner wave)
        constexpr int NG = BK / 8 / WK;                          // fragment groups per wave and K-tile
        float4 a4[NG][T::kTM], b4[NG][T::kTN];                   // every fragment of this K-tile, read up front
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int kg = WK > 1 ? wk * 8 : g * 8;              // first k of the group
            if (ABL == 7 || ABL >= 8) {                           // ablation: no LDS reads in the loop (values made up from the lane id)
#pragma unroll
                for (int i = 0; i < T::kTM; ++i) a4[g][i] = make_float4(abl_frag, abl_frag + 1.f, abl_frag + 2.f, abl_frag + 3.f);
#pragma unroll
                for (int j = 0; j < T::kTN; ++j) b4[g][j] = make_float4(abl_frag, abl_frag - 1.f, abl_frag - 2.f, abl_frag - 3.f);
                continue;
            }
#pragma unroll
            for (int i = 0; i < T::kTM; ++i)
                a4[g][i] = *reinterpret_cast<const float4*>(&DR_AS(buf)[wm * T::kWTM + i * 32 + li][(kg + lk * 4) ^ T::swz(li)]);
#pragma unroll
            for (int j = 0; j < T::kTN; ++j)
                b4[g][j] = *reinterpret_cast<const float4*>(&DR_BS(buf)[wn * T::kWTN + j * 32 + li][(kg + lk * 4) ^ T::swz(li)]);
        }
        if constexpr (BF) {
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int i = 0; i < T::kTM; ++i)
#pragma unroll
                    for (int j = 0; j < T::kTN; ++j)
                        accp[g % KACC][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(dr_bf16x8, a4[g][i]), __builtin_bit_cast(dr_bf16x8, b4[g][j]), accp[g % KACC][i][j], 0, 0, 0);
        } else {
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int i = 0; i < T::kTM; ++i)
#pragma unroll
                    for (int j = 0; j < T::kTN; ++j) {
                        const float av = s4 == 0 ? a4[g][i].x : s4 == 1 ? a4[g][i].y : s4 == 2 ? a4[g][i].z : a4[g][i].w;
                        const float bv = s4 == 0 ? b4[g][j].x : s4 == 1 ? b4[g][j].y : s4 == 2 ? b4[g][j].z : b4[g][j].w;
                        const int q = (g * 4 + s4) % KACC;                  // compile-time after unrolling
                        if (ABL == 2) accp[q][i][j][0] = fmaf(av, bv, accp[q][i][j][0]);
                        else accp[q][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accp[q][i][j], 0, 0, 0);
                    }
        }
        if (ABL == 4) {
            if (more) abl_sink += a_reg[0].x + (BK == 16 ? b_reg0.x + (T::kBIters > 1 ? b_reg1.x : 0.f) : b_fat0.x);
        } else if (more) {
            if constexpr (!GL) store_tile(buf ^ 1, was_tail);
        }
        if (ABL != 6 && ABL < 8) __syncthreads();                         // ABL 9 = 8 + no epilogue stores
        }
    };
    // Pairs of K-tiles run unconditionally (a K-tile under "if (t < T_total)" made hipcc carry the accumulators
    // in VGPRs and copy all of them to and from the AGPRs around every MFMA block); an odd last tile follows.
    const int T_pairs = T_total & ~1;
    for (int t = 0; t < T_pairs; t += 2) {
        k_tile(0, true);
        k_tile(1, t + 2 < T_total);
    }
    if (T_total & 1) k_tile(0, false);
    AccT acc[T::kTM][T::kTN];
#pragma unroll
    for (int i = 0; i < T::kTM; ++i)
#pragma unroll
        for (int j = 0; j < T::kTN; ++j) {
            acc[i][j] = accp[0][i][j];
#pragma unroll
            for (int q = 1; q < KACC; ++q) acc[i][j] += accp[q][i][j];
        }
    if (ABL == 4 && abl_sink == 12345.678f) p.y[0] = abl_sink;
    if constexpr (WK > 1) {
        // sum the two K halves: tile by tile, wave wk = 1 parks an accumulator tile in LDS (the operand tiles are dead: the K
        // loop ended on a barrier; 4 KB per wave = one stage of A for both of them), its partner adds it in a fixed order
        static_assert(sizeof(As0) + sizeof(As1) >= WM * WN * 16 * 64 * sizeof(float) && sizeof(As0) == sizeof(As1), "K-split scratch");
        float* park = (wave % (WM * WN)) == 0 ? &As0[0][0] : &As1[0][0];
        static_assert(WM * WN == 2, "one parking buffer per wave pair");
#pragma unroll
        for (int j = 0; j < T::kTN; ++j) {
            if (wk == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) park[r * 64 + lane] = acc[0][j][r];
            }
            __syncthreads();
            if (wk == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][j][r] += park[r * 64 + lane];
            }
            __syncthreads();
        }
        static_assert(T::kTM == 1, "K-split tiles are one MFMA tile tall");
    }

    // ---- epilogue: conv_epilogue.inc -------------------------------------------------------------
    double s1[T::kTN], s2[T::kTN];
#pragma unroll
    for (int j = 0; j < T::kTN; ++j) s1[j] = s2[j] = 0.0;
    constexpr int EP_TM = T::kTM, EP_TN = T::kTN;
    const int ep_m0 = m0 + wm * T::kWTM, ep_n0 = n0 + wn * T::kWTN;
    const unsigned ep_rows = (WK > 1 && wk != 0) ? 0u : 0xFFFFu;       // K-split: the wk = 0 wave holds the sums
    // one batch of 16 rows where the register budget allows it (three or two waves per SIMD), two of 8 in the five-wave kernels (16
    // spilled there: 60-72 bytes of scratch per lane)
    constexpr int EP_BATCH_ROWS = MF == 16 ? 4 : (BM * BN >= 128 * 128 || BK_ == 64) ? 16 : 8;
    constexpr int EP_TS = MF, EP_NR = NR;
    const int ep_lg = MF == 16 ? (lane >> 4) : lk, ep_lc = MF == 16 ? (lane & 15) : li;
    if ((XB & 2) != 0 && p.bst_raw_bf16) {                   // the XB = 2 / 3 kernels: one copy of the epilogue per storage case (conv_epilogue.inc)
        constexpr bool EP_Y16 = false, EP_B16 = true, EP_B16_CONST = true;
#include "conv_epilogue.inc"
    } else if ((XB & 2) != 0 && p.y_bf16) {
        constexpr bool EP_Y16 = true, EP_B16 = false, EP_B16_CONST = false;
#include "conv_epilogue.inc"
    } else {
        constexpr bool EP_Y16 = false, EP_B16 = false, EP_B16_CONST = false;
#include "conv_epilogue.inc"
    }
    if (p.stat_part) {
        // wave partials -> LDS (the operand tiles are dead: the K loop ended on a barrier) -> one row per workgroup
        // [2][WM][BN] doubles: in the first A stage (K-split and 16-column tiles: wider than tall, in the first B stage)
        double* red = reinterpret_cast<double*>((WK > 1 || MF == 16) ? &Bs0[0][0] : &As0[0][0]);
        static_assert(((WK > 1 || MF == 16) ? sizeof(Bs0) : sizeof(As0)) >= sizeof(double) * 2 * WM * BN, "stat scratch does not fit the operand tile");
#pragma unroll
        for (int j = 0; j < T::kTN; ++j) {
            double a = s1[j], b = s2[j];
            if constexpr (MF == 16) {                                       // four lane groups hold four rows each of column lane & 15
                a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
            }
            a += __shfl_xor(a, 32);
            b += __shfl_xor(b, 32);
            if (ep_lg == 0 && wk == 0) {                                    // (K-split: the partner wave accumulated nothing)
                const int col = wn * T::kWTN + j * MF + ep_lc;
                red[(0 * WM + wm) * BN + col] = a;
                red[(1 * WM + wm) * BN + col] = b;
            }
        }
        __syncthreads();
        for (int e = tid; e < 2 * BN; e += T::kThreads) {                   // (2 * BN > 256 on the 160-column tile)
            const int which = e / BN, col = e % BN, n = n0 + col;
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < WM; ++w) t += red[(which * WM + w) * BN + col];
            if (n < p.Cout) p.stat_part[((long)which * p.Cout + n) * gx + mblk] = t;
        }
    }
}

#undef DR_AS
#undef DR_BS

// Host-side launcher: picks the tile shape from (M, Cout).
int launch_conv_igemm(const ConvParams& p, hipStream_t stream);

}  // namespace dr
