/* densereg_debug.h -- test / micro-benchmark hooks.  NOT part of the integration surface and NOT in the product library:
 * libdensereg_hip_dbg.so (the product's sources built with -DDR_DEBUG_HOOKS) exports them so that tests/ and tools/ can drive
 * a single kernel through a C entry point with raw pointers.  The product library libdensereg_hip.so exports densereg.h and
 * densereg_profile.h only. */
#ifndef DENSEREG_DEBUG_H_
#define DENSEREG_DEBUG_H_
#include "densereg.h"
#ifdef __cplusplus
extern "C" {
#endif

/* One stride-1 SAME convolution through the implicit-GEMM MFMA kernel.
 * x: (B,H,W,*) with channel stride x_cs (multiple of 4) ; w: HWIO (k,k,Cin,Cout) ; y: channel stride y_cs.
 * v = conv*scale[n] + shift[n] (NULL = 1 / 0); relu; + res (stride res_cs) ; rows with rowmask[m] < thresh
 * read as zero; stat (2*Cout doubles, pre-zeroed): per-channel sum / sum of squares of the raw conv.
 * All pointers are device pointers.  Synchronises the stream. */
int dr_dbg_conv2d(int B, int H, int W, int Cin, int Cout, int k, const float* x, int x_cs, const float* w,
                  const float* scale, const float* shift, int relu, const float* res, int res_cs,
                  const float* rowmask, float thresh, float* y, int y_cs, double* stat, dr_stream stream);

/* Weight gradient of one conv on caller buffers (device pointers, NHWC with row strides x_cs / g_cs):
 * dw[k][k][Cin][Cout] = sum over pixels of x(shifted by the tap, zero outside the image or where
 * rowmask[pixel] < thresh) * g.  T = 64 or 128 picks the channel tile (96: the kernel-row variant; 160 + id: a 16x16-tile kernel of
 * conv_wgrad16.h -- 161 / 162 need k = 3, 163..165 k = 1), nsplit the pixel-axis slabs. */
int dr_dbg_wgrad(int B, int H, int W, int Cin, int Cout, int k, const float* x, int x_cs, const float* g, int g_cs,
                 const float* rowmask, float thresh, int T, int nsplit, float* dw, dr_stream stream);

/* Micro-benchmark of the weight-gradient kernel + slab fold: microseconds per call for channel tile T (0 = planner's, 160 + id as above)
 * and nsplit slabs (0 = planner's; *nsplit_used reports the count actually run). */
int dr_dbg_wgrad_bench(int B, int H, int W, int Cin, int Cout, int k, int T, int nsplit, int iters, float* us_out,
                       int* nsplit_used);

/* Micro-benchmark of the BatchReNorm streaming kernels on an [M][C] tensor: us_out[3] = microseconds per launch
 * of (train apply, backward reduce, backward apply).  reduce_blocks > 0 overrides the backward-reduce grid. */
int dr_dbg_bn_bench(long M, int C, int reduce_blocks, int iters, float* us_out);

/* ONE conv -> BatchReNorm(train) [-> ReLU] [+ residual] layer, forward and backward, through exactly the kernels and the
 * launch logic the executors use (train_exec.inc: run_conv_train / backward_conv), on caller buffers -- so the train-mode
 * BatchReNorm kernels can be pinned against an fp64 autograd of the same layer (network/slim/ops.py:130-171).
 * Tensors are NHWC with channel stride cs = round_up(Cout, 4) unless a stride is given.  Backward seed: either `dout`
 * (the layer runs its own reduce pass) or a consumer convolution (`gr`, `wr`: kr x kr, Cout -> Cr): then dOut is
 * produced by that consumer's dgrad launch, which also computes the layer's backward sums in its epilogue (the
 * "single reader" path of plan_backward); with BOTH given, `dout` is what other readers already accumulated and the
 * consumer's dgrad adds to it as the last writer of the gradient buffer (the "last writer" path).  All pointers are
 * device pointers; synchronises the stream. */
typedef struct dr_dbg_bn_args {
    int B, H, W, Cin, Cout, k;
    const float* x; int x_cs;                 /* layer input */
    const float* w;                           /* HWIO (k,k,Cin,Cout) */
    const float* gamma; const float* beta; const float* mm; const float* mv;      /* [Cout]; mm/mv = moving stats before */
    float r_max, d_max; int relu;
    const float* res;                         /* nullable: residual added after the activation, [M][cs] */
    const float* dout;                        /* [M][cs], or NULL when a consumer is given */
    const float* gr; int gr_cs; const float* wr; int kr; int Cr;
    float* y; float* raw;                     /* out: activation and raw conv output, [M][cs] */
    float* bnc;                               /* out: [4][Cout] mean, inv_std, r, d of this step */
    float* mm_next; float* mv_next;           /* out: moving stats after (zero-debiased first update) */
    float* dout_used;                         /* out: the dOut the backward saw, [M][cs] */
    float* draw;                              /* out: gradient wrt the raw conv output, [M][cs] */
    float* dgamma; float* dbeta;              /* out: [Cout] */
    float* dres;                              /* out, nullable: gradient of the residual input, [M][cs] */
    int fwd_rows, bwd_rows;                   /* out: partial rows of the two reductions (which finalize path ran) */
} dr_dbg_bn_args;
int dr_dbg_bn_layer(dr_dbg_bn_args* a, dr_stream stream);

/* The backward of a BIAS conv with ReLU (and dropout) as its single reader's dgrad produces it (conv_igemm.h, bst_act):
 * `out` [M][cs] is the layer's forward output, the reader is a kr x kr conv C -> Cr with output gradient `gr`; writes
 * g [M][cs] = dOut * factor * [out > 0] (dOut = the reader's input gradient) and dbias[C] += column sums of g. */
int dr_dbg_act_dgrad(int B, int H, int W, int C, int Cr, int kr, const float* out, const float* gr, int gr_cs, const float* wr,
                     float factor, float* g, float* dbias, dr_stream stream);

/* Max-pool k x k / stride 2, TF 'SAME' (ops.max_pool, network/slim/ops.py:640-669) on dense [B][H][W][C] device buffers,
 * C % 4 == 0: y [B][ceil(H/2)][ceil(W/2)][C] and, from the arg-max the forward records, dx = (acc ? dx : 0) + dy routed to
 * the FIRST maximum of every window in scan order (the convention of the oracle's autograd). */
int dr_dbg_maxpool(int B, int H, int W, int C, int k, const float* x, float* y, const float* dy, float* dx, int acc,
                   dr_stream stream);

/* Micro-benchmark one conv shape on self-allocated buffers: average milliseconds per launch.
 * tile = -1 (heuristic) or a tile id (0 128x128, 1 64x128, 2 128x64, 3 64x64, 4 128x32);
 * abl = 0 product kernel, 1/2/3 = ablations of the 128x128 kernel (no refills / no MFMA / no stores),
 * 5 = product kernel with the fused residual add, 6 = product kernel on all-zero operands,
 * 7 = 64x128 kernel without refills. */
int dr_dbg_conv_bench(int B, int H, int W, int Cin, int Cout, int k, int tile, int abl, int iters, float* ms_out);

/* Sustained fp32 MFMA TFLOP/s of the device without memory traffic (register-only chains of
 * v_mfma_f32_32x32x2_f32; `waves_per_simd` resident waves; zero_data = 1 feeds zeros, the DVFS best case). */
int dr_dbg_mfma_peak(int iters, int waves_per_simd, int zero_data, float* tflops_out);

/* While on (process-global), dr_dbg_conv2d and dr_dbg_conv_bench (abl 0) pack their weights as bf16 and run the
 * bf16 matrix-core variant of the tile (dr_set_precision(DR_PREC_BF16) on a handle). */
int dr_dbg_force_bf16(int on);

/* While on (process-global, with dr_dbg_force_bf16(1)): the bf16-STORAGE variants of the bf16 matrix-core kernels --
 * dr_dbg_conv2d reads x as bf16 elements (x_cs = element stride), dr_dbg_wgrad reads g as bf16 elements, dr_dbg_bn_layer
 * writes draw as bf16 elements (the executor stores a BatchReNorm layer's dRaw that way on the bf16 path). */
int dr_dbg_force_bf16_storage(int on);
/* conv_x3.h (fp32-accurate products on the bf16 matrix cores): -1 = DR_CONV_X3 / the measured rule, 0 = never, 1 = the rule, 2 = wherever
 * the kernel can run -- for the debug entries and every handle of the process */
int dr_dbg_force_x3(int mode);
/* ... 3 / 4 / 5: as 2 with one accumulator / the three-stage LDS ring / four waves per workgroup; 6: as 2, and the debug entries store the
 * input of a convolution as its three bf16 planes ("P3", densereg_amd/csrc/conv_p3.h) so that conv_p3_kernel runs where its tile applies.
 * dr_dbg_p3_launches: conv_p3_kernel launches of this process so far (a test's proof that the kernel under test is the one that ran). */
long dr_dbg_p3_launches(void);
/* ... 7: as 2, but conv_x3_kernel also where conv_x3h_kernel (densereg_amd/csrc/conv_x3h.h: 3x3 layers with the haloed input tile resident in
 * LDS across the nine taps) would run.  dr_dbg_x3h_launches: conv_x3h_kernel launches of this process so far. */
long dr_dbg_x3h_launches(void);

/* Partial statistics rows one wave of a BatchReNorm finalize launch folds before the micro-batch group gets another wave
 * (densereg_amd/csrc/train_kernels.h: bn_finalize_split; default 512, 0 restores it).  Process-global; tests lower it so that
 * small layers run the several-waves-per-group path. */
int dr_dbg_bn_finalize_rows(int rows);

/* Force the conv tile of every following launch (-1 = heuristic; ids as in dr_dbg_conv_bench).
 * Process-global; tests use it to check every tile shape against the reference. */
int dr_dbg_force_tile(int tile);

#ifdef __cplusplus
}
#endif
#endif
