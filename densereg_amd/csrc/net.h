// net.h -- the runtime around the kernels: graph builder for network/um_v1.py, parameter registry
// (TF variable names), buffer plan, and the forward/backward executors.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../include/densereg.h"
#include "dr_platform.h"
#include "conv_wgrad.h"
#include "conv_wgrad16.h"
#include "hg_fused.h"
#include "kernels_misc.h"

namespace dr {

struct Tensor {
    int id = 0;
    int H = 0, W = 0, C = 0, cs = 0;   // cs = channel stride (C rounded up to 4)
    float* p = nullptr;                // forward values   [max_batch*H*W*cs]
    float* g = nullptr;                // gradient buffer (training handles)
    size_t off = 0;                    // p = <activation arena> + off, g = <gradient arena> + off: re-bound per micro-step slot (StepSlot)
    bool needs_grad = true;
    bool needs_zero = false;           // gradient buffer must be zeroed before a backward pass (see plan_backward)
    int grad_C = 0;                    // channels [0, grad_C) of the gradient have a consumer (plan_backward): the tail of a
                                       // concat buffer written by an op without a backward (the uvd planes) needs no dgrad
    std::string tag;
    // bf16 matrix-core training: an activation whose ONLY reader is a conv (its A operand and, in the backward sweep, that conv's
    // weight-gradient x operand) may be stored as bf16 by its BatchReNorm apply pass -- both readers round it to bf16 while staging
    // anyway.  reader_op: that conv's index in dr_handle::ops (plan_backward), -1 = not eligible; is_bf16: what the buffer
    // holds after the forward in progress.
    int reader_op = -1;
    bool is_bf16 = false;
    bool g_bf16 = false;                // the GRADIENT buffer holds bf16 elements this backward sweep (bf16 path: written by the only
                                        // reader's input-gradient launch, read by the producer's BatchReNorm backward apply; train_exec.inc)
};

struct TView {
    Tensor* t = nullptr;
    int coff = 0;
    int C = 0;
    bool valid() const { return t != nullptr; }
    View fwd() const { return View{t->p, t->cs, coff, C}; }
    View grad() const { return View{t->g, t->cs, coff, C}; }
};

struct ConvLayer {
    std::string name;                  // TF scope: 'hg_imgproc/Conv_3', 'Conv_40'
    int k = 1, stride = 1, cin = 0, cout = 0;
    bool bn = false, relu = false;
    float wd = 0.f;
    int H = 0, W = 0;                  // output spatial dims
    // offsets (in floats) into the flat trainable buffer
    size_t w_off = 0, beta_off = 0, gamma_off = 0, bias_off = 0;
    // offsets into the flat state buffer (moving_mean, moving_variance)
    size_t mm_off = 0, mv_off = 0;
    float r_max = 1.f, d_max = 0.f, curr_t = 0.f;      // BatchReNorm schedule scalars (ops.py:114-128)
    // zero-debias shadow of assign_moving_average [TF1.3-semantics]: biased accumulators + step
    size_t shadow_off = 0;             // 2*cout floats in the shadow buffer (biased mean, biased var)
    int shadow_step = 0;
    // packed weights
    int Kp = 0, Np = 0;                // forward:  [taps][Kp][Np]
    int KpT = 0, NpT = 0;              // dgrad:    [taps][KpT][NpT]  (cout -> K, cin -> N, taps flipped)
    size_t wp_off = 0, wpT_off = 0;    // offsets into the packed buffers
    size_t fold_off = 0;               // 2*cout floats: scale, shift (eval fold or train step values)
    size_t bnc_off = 0;                // 4*cout floats: mean, inv_std, r, d saved by the train forward
    Tensor* raw = nullptr;             // pre-BN conv output (training)
    int bst_rows = 0;                  // backward sweep: > 0 = the consumer's dgrad already wrote this many partial rows of
                                       // this layer's BatchReNorm backward sums into stat_part2 (train_exec.inc)
    int ep_fwd = 0, ep_bwd = 0;        // launches so far of the look-back apply kernels (targets of the hand-off counters)
    float* g_keep = nullptr;           // small layers (training): a private dRaw buffer that outlives the layer's step of the
                                       // backward sweep, so its weight gradient can run in the grouped launch at the end
    size_t g_keep_off = (size_t)-1;    // g_keep = <slot's g_keep arena> + g_keep_off ((size_t)-1: none)
};

enum OpKind { OP_STEM, OP_CONV, OP_POOL, OP_UPADD, OP_UVD, OP_COPY, OP_FORK, OP_JOIN };

// Executor lanes.  The two branches of an hourglass level (um_v1.py:51-69: the residual at this resolution and
// the pooled pyramid below it) are independent until their upsample-add, and everything below 32x32 is a chain of
// tiny, latency-bound launches.  The graph builder puts hourglass depth d on lane d and its lower pyramid on lane
// d+1; lane 0 is the caller's stream, lanes >= 1 are library-owned HIP streams.  OP_FORK (parent -> child) and
// OP_JOIN (child -> parent) are event edges; the reverse sweep of the backward pass swaps their roles.
constexpr int DR_MAX_LANES = 8;

struct Op {
    OpKind kind;
    TView in, in2, out;                // in2: residual source (conv) / low-res input (upadd)
    int conv = -1;                     // ConvLayer index
    bool masked = false;               // depth mask on the input rows (um_v1.py:146-148)
    int dropout = -1;                  // dropout slot index (stack*2 + i) or -1
    int pool_k = 0;
    unsigned char* pool_arg = nullptr; // training: arg-max position per pooled element, [B*Ho*Wo][C] bytes (maxpool_kernel)
    size_t pool_off = (size_t)-1;      // pool_arg = <slot's arg-max arena> + pool_off
    int lane = 0;                      // stream lane the op runs on (FORK/JOIN: the parent lane)
    int lane2 = 0;                     // FORK/JOIN: the child lane
    int ev = -1;                       // FORK/JOIN: index into dr_handle::lane_ev
    TView uvd0, uvd1;                  // OP_UVD destinations
    bool ow_in = false, ow_in2 = false; // backward: this op is the FIRST writer of grad(in) / grad(in2) -> overwrite
    int bst_conv = -1;                 // backward: this conv's dgrad also reduces the BatchReNorm backward sums of conv #bst_conv
    bool bst_last = false;             // ... as the LAST (accumulating) writer of that layer's output gradient rather than the only one
};

enum ParamKind { PK_WEIGHT, PK_BETA, PK_GAMMA, PK_BIAS, PK_MMEAN, PK_MVAR, PK_RMAX, PK_DMAX, PK_CURRT };

struct ParamInfo {
    std::string name;
    int32_t dims[4] = {0, 0, 0, 0};
    int32_t ndim = 0;
    bool trainable = false;
    ParamKind kind = PK_WEIGHT;
    int conv = -1;
    size_t count = 0;
};

enum KernelId {
    KID_CONV_128x128, KID_CONV_64x128, KID_CONV_128x64, KID_CONV_64x64, KID_CONV_128x32, KID_CONV_64x64_K64, KID_CONV_SPLITK, KID_CONV_64x96, KID_CONV_64x160, KID_CONV16_64x80, KID_CONV16_64x144, KID_CONV16_64x160, KID_STEM, KID_POOL, KID_UPADD, KID_UVD, KID_COPY,
    KID_HG_FUSED, KID_VOTE, KID_BN, KID_WGRAD, KID_WGRAD_FOLD, KID_ELTWISE, KID_LOSS, KID_ADAM,
    KID_WGRAD_128, KID_WGRAD_64, KID_WGRAD_ROW, KID_WGRAD_GROUP, KID_WGRAD_16,   // one row per weight-gradient kernel template (KID_WGRAD: the stem's)
    KID_CONV_X3,                     // conv_x3.h (named conv_x3_128x128: bench.py prices it against the bf16 matrix cores / 6)
    KID_WGRAD_X3_128, KID_WGRAD_X3_64,   // conv_wgrad_x3.h
    KID_CONV_P3,                     // conv_p3.h (the x3 products on a P3-stored input; priced like conv_x3_128x128)
    KID_COUNT
};
static const char* const kKernelNames[KID_COUNT] = {
    "conv_igemm_128x128", "conv_igemm_64x128", "conv_igemm_128x64", "conv_igemm_64x64", "conv_igemm_128x32", "conv_igemm_64x64k64", "conv_splitk_32x32",
    "conv_igemm_64x96", "conv_igemm_64x160", "conv_igemm16_64x80", "conv_igemm16_64x144", "conv_igemm16_64x160",
    "stem_conv", "maxpool",
    "upsample_add", "uvd", "copy_channels", "hourglass_tail_fused", "vote", "batch_renorm", "stem_wgrad", "wgrad_fold", "eltwise_bwd", "loss", "adam",
    "conv_wgrad_128", "conv_wgrad_64", "conv_wgrad_row96", "conv_wgrad_group", "conv_wgrad16", "conv_x3_128x128", "conv_wgrad_x3_128", "conv_wgrad_x3_64", "conv_p3_128x128"};

// The part of one hourglass below 16x16 pixels (hg_fused.h): in eval mode the ops [first_op, last_op] and the pool at pool_op are
// ONE launch.  conv[]: the ConvLayer indices of its eight residual modules in execution order.
struct FusedRegion { int pool_op = -1, first_op = -1, last_op = -1; TView in, out; int conv[24]; };

struct ProfRecord { rt::Event a, b; int kid; int tag; double flops; double bytes; };
struct RegSeg;

// Everything ONE training micro-step in flight owns: what its forward writes and its backward reads (activations, raw conv outputs,
// BatchReNorm step values, arg-max bytes, the depth mask, copies of the caller's inputs), the backward sweep's scratch, and a gradient
// accumulator.  A handle has one slot -- or two (dr_set_pipeline(h, 2)): consecutive micro-steps then alternate between the slots, each
// on its slot's stream, and the kernels of micro-step k+1 fill the launch boundaries and small-grid chains of micro-step k (measured:
// two independent engines on one MI355X give 2392 crops/s together against 2047 for one, profiles/r03_experiments.md).  The executors
// keep reading the handle's working fields (act_arena, fold, tiny, ... and every Tensor::p / g): bind_slot() points them at a slot.
struct StepSlot {
    float* act_arena = nullptr; float* grad_arena = nullptr; float* fold = nullptr; float* bnc = nullptr; float* tiny = nullptr;
    float* g_keep_arena = nullptr; unsigned char* pool_arg_arena = nullptr;
    // per executor lane (Op::lane indexes these whether or not the lanes run on their own streams): dRaw scratch, the partial-sum
    // rows of the BatchReNorm reductions (own layer / written by a dgrad for the next layer), backward coefficients, wgrad slabs
    float* scratch_l[DR_MAX_LANES] = {}; double* stat_part_l[DR_MAX_LANES] = {}; double* stat_part2_l[DR_MAX_LANES] = {};
    float* bn_coef_l[DR_MAX_LANES] = {}; float* wg_partial_l[DR_MAX_LANES] = {};
    double* loss_acc = nullptr; float* gacc = nullptr;
    float* dm_copy = nullptr; float* aux_copy = nullptr;     // the caller's crops / poses | camera | centres of mass of this micro-step
    WgradGroupSeg* group_dev = nullptr; std::vector<WgradGroupSeg> group_uploaded;
    FoldSeg* fold_dev = nullptr; std::vector<FoldSeg> fold_uploaded;       // (a slot's table may be re-planned while the other's fold is queued)
    hipStream_t stream = nullptr;                            // library-owned (two slots), else the caller's stream is used
    hipStream_t wg_stream = nullptr; rt::Event wg_ready{}, wg_done{};
    rt::Event in_ev{}, fwd_done{}, loss_done{}, step_done{};
    bool fwd_recorded = false, step_recorded = false, loss_recorded = false;   // the events above carry a record
    bool events_live = false;                                // in_ev / fwd_done / loss_done / step_done exist (created with the slot streams)
    bool grads_dirty = false;                                // gacc holds contributions not yet folded into slot 0's (dr_sync_grads)
    bool owns = false;                                       // slot 1: buffers allocated by dr_set_pipeline (slot 0 aliases the handle's)
};

}  // namespace dr

struct dr_handle {
    dr_config cfg{};
    mutable std::string err;
    std::vector<std::unique_ptr<dr::Tensor>> tensors;
    std::vector<dr::ConvLayer> convs;
    std::vector<dr::Op> ops;
    std::vector<dr::ParamInfo> params;
    int map_hw = 0;

    // device buffers
    float* flat_param = nullptr; size_t n_train = 0;
    float* flat_grad = nullptr;  float* adam_m = nullptr; float* adam_v = nullptr;
    float* flat_state = nullptr; size_t n_state = 0;       // moving stats
    float* flat_state_next = nullptr;                      // written by a training forward, then swapped in
    float* shadow = nullptr;     size_t n_shadow = 0;      // zero-debias biased accumulators
    float* wp = nullptr;         size_t n_wp = 0;          // packed forward weights
    float* wpT = nullptr;        size_t n_wpT = 0;         // packed dgrad weights
    __bf16* wp3 = nullptr;       __bf16* wp3T = nullptr;   // conv_x3.h: the same two buffers as three bf16 planes (3 * n_wp / 3 * n_wpT elements)
    float* fold = nullptr;       size_t n_fold = 0;        // per-BN-layer scale|shift
    float* bnc = nullptr;        size_t n_bnc = 0;
    float* act_arena = nullptr;  size_t n_act = 0;
    float* grad_arena = nullptr; size_t n_gact = 0;
    float* scratch = nullptr;    size_t n_scratch = 0;     // dRaw scratch (training) / dense copies
    float* scratch_l[dr::DR_MAX_LANES] = {};                // one per lane (index 0 aliases `scratch`)
    hipStream_t lane_stream[dr::DR_MAX_LANES] = {};         // [0] unused: lane 0 is the caller's stream
    std::vector<dr::rt::Event> lane_ev;                     // one ordering event per FORK / JOIN op
    int n_lanes = 1;
    // executable graphs of the inference entry points, keyed by (entry, B, caller pointers); opt-in with DR_GRAPHS=1
    struct GraphEntry { int entry; int B; const void* ptr[6]; dr::rt::Graph g; };
    std::vector<GraphEntry> graphs;
    hipStream_t cap_stream = nullptr;                      // library-owned stream the captures are recorded on
    bool use_graphs = false;
    int precision = 0;                                      // dr_set_precision: 0 = fp32 MFMA, 1 = bf16 MFMA (inference handles)
    std::vector<dr::FusedRegion> fused;                     // one per stack (graph builder); used by the eval forward when fuse_tail
    bool fuse_tail = true;                                  // dr_set_fusion / DR_FUSE_TAIL=0: every op of the hourglass bottoms launches on its own
    bool last_eval_fused = false;                           // the last eval forward skipped the fused regions' intermediate tensors
    bool fuse_bn_bwd = true;                               // DR_FUSE_BN_BWD=0: every BatchReNorm layer runs its own reduce pass
    bool multi_stream = false;                             // DR_MULTI_STREAM=1 turns the lanes on; off or profiling: every lane = caller's stream
    float* tiny = nullptr;                                  // (B,h,w) normalised depth at map resolution
    float* tiny_ext = nullptr;                              // same, for dr_vote on external maps
    float* zeros = nullptr;                                 // 256 B of zeros: target of predicated-off loads
    float* losses = nullptr;                                // 4 floats (device)
    const float* dm_in = nullptr;                           // input of the last forward (not owned)

    // network outputs per stack (dense-ish tensors)
    std::vector<dr::Tensor*> hm, hm3, um;
    dr::Tensor* input = nullptr;       // not allocated: points at caller's dm
    bool finalized = false;
    bool last_forward_train = false;
    int last_B = 0;
    int dropout_mode = 0; const uint8_t* keep_mask = nullptr; uint64_t seed = 0;
    double flops_per_crop = 0.0;
    bool profiling = false;
    int prof_tag = -1;                                     // conv index of the op being executed (profiling detail)
    std::vector<dr::ProfRecord> prof;
    // training-only state
    dr::RegSeg* reg_segs = nullptr; int n_reg = 0;        // weight segments with weight_decay > 0
    double* loss_acc = nullptr;                            // 4 doubles: hm, hm3, um, reg
    float* bn_coef = nullptr;                              // 3*max(cout) floats (BatchReNorm backward)
    float* bn_coef_l[dr::DR_MAX_LANES] = {};                // one per lane (index 0 aliases `bn_coef`)
    double* stat_part_l[dr::DR_MAX_LANES] = {};             // per lane: partial-sum rows of the BatchReNorm reductions
    double* stat_part2_l[dr::DR_MAX_LANES] = {};            // per lane: rows written by a dgrad for the NEXT layer's backward
    size_t n_stat_part = 0;
    float* wg_partial = nullptr; size_t n_wg_partial = 0;  // split-K slabs of the weight gradient
    float* wg_partial_l[dr::DR_MAX_LANES] = {};             // one per lane (index 0 aliases `wg_partial`)
    // deferred slab fold (single-stream executor): every layer keeps its slabs until one fold launch at the end
    std::vector<dr::FoldSeg> fold_host;                     // segments of the sweep in progress
    void* zero_dev = nullptr; int zero_nseg = 0;            // ZeroSeg table of the gradient buffers dr_loss clears (plan_backward)
    dr::FoldSeg* fold_dev = nullptr;                        // device copy (capacity = number of convs)
    size_t fold_head = 0;                                  // floats at the start of wg_partial kept for immediate folds
    size_t fold_used = 0;                                  // floats of wg_partial handed out in this sweep
    int fold_blocks = 0;
    size_t n_loss_part = 0;                                // rows of the loss kernel's partial sums (loss_acc)
    void* pack_dev = nullptr; int pack_nseg = 0, pack_blocks = 0;   // segment table of the one-launch weight packing
    // grouped weight gradient of the small layers (conv_wgrad.h: conv_wgrad_group_kernel), single-stream executor
    unsigned char* pool_arg_arena = nullptr;                // max-pool arg-max bytes (Op::pool_arg)
    float* g_keep_arena = nullptr;                          // the layers' private dRaw buffers
    std::vector<dr::WgradGroupSeg> group_host;              // segments of the sweep in progress (what group_dev holds: StepSlot::group_uploaded)
    dr::WgradGroupSeg* group_dev = nullptr;
    int group_blocks = 0; double group_flops = 0, group_bytes = 0;
    bool group_wgrad = true;                                // DR_GROUP_WGRAD=0: every layer launches its own weight gradient
    // Weight gradients of the full-resolution layers on a library-owned low-priority stream (default; DR_WGRAD_STREAM=0 off): they are
    // needed only by the slab fold at the end of the sweep, so they are queued while the sweep runs the heads of a stack and
    // released when it enters the hourglass below -- a ~1.4 ms chain of launches too small to fill the chip.
    struct PendingWgrad { dr::WgradParams p; int kind; int grid; };   // kind: 0 <64>, 1 <128>, 2 row kernel, 3 / 4 bf16 <64> / <128>, 5 conv_wgrad_tail_kernel, 16 + id: conv_wgrad16.h
    std::vector<PendingWgrad> wg_pending;
    hipStream_t wg_stream = nullptr;
    dr::rt::Event wg_ready{}, wg_done{};
    bool wgrad_stream = false, wg_side_used = false;
    int wg_flush_every = 0;                                 // > 0: release the queue after every n-th queued layer as well (DR_WGRAD_STREAM=<n+1>)
    int* bn_flags = nullptr;                                // [2 counters + 2 expiry flags] per conv: look-back hand-off of the BatchReNorm
                                                            // coefficients (train_kernels.h), opt-in with DR_BN_LOOKBACK=1 (measured slower)
    bool bn_lookback = false;
    bool bf16_act = true;                                   // DR_BF16_ACT=0: single-conv-reader activations stay fp32 on the bf16 path
    bool bf16_draw = true;                                  // DR_BF16_DRAW=0: dRaw stays fp32 on the bf16 matrix-core path (train_exec.inc)
    bool bf16_gact = true;                                  // DR_BF16_GACT=0: the gradient of a single-conv-reader activation stays fp32 on that path
    bool bf16_raw = true;                                   // DR_BF16_RAW=0: the raw outputs of BatchReNorm convs stay fp32 on that path (train_exec.inc)
    bool fold_is_eval = false;                             // `fold` holds the eval-mode BN fold
    const float* dm_train = nullptr;                       // input of the last dr_forward_train
    // micro-step slots (StepSlot above)
    dr::StepSlot slot[2];
    int groups = 1;                                        // micro-batch groups per call (dr_set_groups): 1 = the batch is one micro-batch
    int pipe_depth = 1;                                    // 1: every call runs on the caller's stream; 2: two micro-steps in flight
    int cur_slot = 0, next_slot = 0;                       // slot of the micro-step in progress / of the next dr_forward_train
    float* gacc = nullptr;                                 // gradient accumulator of the bound slot (slot 0's IS flat_grad)
    size_t n_keep = 0, n_pool_arg = 0, n_loss_acc = 0, n_group_dev = 0;   // sizes a second slot needs (elements / bytes / doubles / segments)
};
