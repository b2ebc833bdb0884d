// train_kernels.h -- training-side kernels: train-mode BatchReNorm, backward passes, loss, Adam.
//
// Reference semantics (paths relative to the reference tree):
//   BatchReNorm train mode      network/slim/ops.py:130-171
//   loss + target synthesis     model/hourglass_um_crop_tiny.py:193-274, 323-371
//   accumulate / clip / Adam    model/train_single_gpu.py:69-89 ; hourglass_um_crop_tiny.py:436-439
// Gradients are derived by hand from those forward definitions (TF would autodiff the same graph);
// tests/ check them against torch autograd through the CPU oracle.
#pragma once
#include "dr_platform.h"
#include "kernels_misc.h"

namespace dr {

// ------------------------------------------------------------------------------------------------
// BatchReNorm, train mode (ops.py:130-171), fused "finalize + normalise":
//   mean/var: biased moments of the raw conv output (fp64 sums from the conv epilogue)
//   r = clip(std/std_mov, 1/r_max, r_max), d = clip((mean-mean_mov)/std_mov, +-d_max)   [stop-gradient]
//   out = relu(((x-mean)*inv_std*r + d)*gamma + beta) + res  ==  relu(x*scale + shift) + res
// Every workgroup derives scale/shift of its channels from the sums (a few flops per channel);
// workgroup 0 also persists what the backward pass and the next step need: scale|shift, bnc =
// (mean, inv_std, r, d) and the NEW moving statistics.  Moving stats follow "read old, then update"
// (SURVEY Appendix C.2): they are read from mm/mv and written to mm_next/mv_next (the host swaps the two
// state buffers after the forward), so no workgroup can observe a half-updated state.
// assign_moving_average(decay=.99) with [TF1.3-semantics] zero_debias=True:
//   biased -= (biased-value)*(1-decay); var = biased/(1-decay^step).
// Mapping: a thread owns 4 consecutive channels (fixed) and strides over rows: float4 traffic, no div/mod
// in the loop, scale/shift live in registers.
// ------------------------------------------------------------------------------------------------
struct BnTrainParams {
    const float* raw; int raw_cs; long M; int C;
    const double* part; int part_rows;          // [2][C][part_rows] partial sums / sums of squares (conv epilogue)
    const float* beta; const float* gamma;
    const float* mm; const float* mv;           // moving stats BEFORE this step
    float* mm_next; float* mv_next;             // moving stats AFTER this step
    float* shadow_mean; float* shadow_var; int shadow_step;   // step AFTER this update (>=1); 0 = plain EMA
    float r_max, d_max, eps, decay;
    float* scale; float* shift;                 // persisted fused multiply-add (backward relu mask)
    float* bnc;                                 // persisted [4][C]: mean, inv_std, r, d
    int relu;
    View res, out;
    int* flag; int flag_target;                 // look-back hand-off (bn_train_apply_kernel<2>): groups published so far / wanted
    int raw_bf16;                               // raw holds bf16 elements (same element stride raw_cs): the conv epilogue stored it that way (ConvParams::y_bf16)
    int out_bf16;                               // out holds bf16 elements (same element stride out.cs; whole channel groups of four,
                                                // no residual): the activation's only readers round it to bf16 while staging
    // Micro-batch groups (dr_set_groups): the M rows are `groups` consecutive micro-batches of Mg rows each -- the reference's
    // gradient-accumulation micro-steps (train_single_gpu.py:138-150) run as ONE pass of launches.  Statistics, r / d and the
    // moving-state update are per micro-batch, in order (group g reads the moving statistics group g-1 left: ops.py:134-162);
    // the partial rows of group g are rows [g*rows_per_group, (g+1)*rows_per_group) of `part`; scale | shift and bnc of group g
    // live fold_stride / bnc_stride floats behind group g-1's.  groups <= 1: one batch, M rows (everything above as it was).
    int groups; int rows_per_group; long Mg; long fold_stride; long bnc_stride;
    int fin_split;                              // finalize launch: waves per group (launch_bn_*_finalize sets it)
    float r_max_g[8], d_max_g[8];               // the schedule scalars of groups 0..groups-1 (r_max / d_max above = group 0's)
};
constexpr int kMaxGroups = 8;

// Look-back hand-off of per-channel coefficients inside ONE launch instead of a separate finalize launch per layer and
// sweep (~200 launches of 5-6 us per training step).  OPT-IN (DR_BN_LOOKBACK=1) and NOT the default: measured on MI355X it
// is slower in both forms tried -- the eight XCDs' L2s are not coherent with each other, so the consumers either read
// every coefficient with agent-scope loads (4 M uncached loads per launch: BatchReNorm 5.5 -> 30 ms per step) or take one
// acquire per workgroup, which drops the conv output the pass is about to stream from the L2 it still sits in
// (5.5 -> 8.1 ms).  A kernel boundary is the cheap way to publish across XCDs on this part.  The first ceil(C/4) workgroups of the grid fold the partial rows of
// "their" four channels (one wave per channel, exactly the finalize kernel's code), write the coefficients, and publish by
// adding to a per-layer counter; every workgroup then waits until the counter reaches its target and reads the coefficients
// with agent-scope loads.  Forward progress: workgroups are dispatched in index order, so every producer is resident (or
// done) before any workgroup that waits for it -- the assumption of every single-pass look-back scan.  The counter is
// monotonic (target = launches so far x groups, kept by the host), so nothing is reset between launches.  The wait is
// bounded: on expiry the workgroup proceeds and raises flag[1] (checked by the tests), it never hangs the device.
constexpr int kBnSpinLimit = 1 << 22;
__device__ __forceinline__ void bn_handoff_publish(int* flag, int groups_done) {
    __threadfence();                                // this thread's coefficient stores: visible device-wide ...
    __syncthreads();                                // ... for every writer of the workgroup ...
    if (threadIdx.x == 0 && groups_done > 0) {
        __threadfence();
        atomicAdd(flag, groups_done);               // ... before the count says so
    }
}
__device__ __forceinline__ void bn_handoff_wait(int* flag, int target) {
    if (threadIdx.x == 0) {
        int spins = 0;
        while ((int)(dr_load_agent_i32(flag) - target) < 0 && ++spins < kBnSpinLimit) dr_spin_pause();
        if (spins >= kBnSpinLimit) atomicAdd(flag + 1, 1);
        // ONE acquire per workgroup, before the barrier releases the other waves: it invalidates the caches all of them read
        // through (this CU's L1, this XCD's L2), so plain loads of the coefficients are fresh afterwards.  (A fence in every
        // wave plus agent-scope loads of every coefficient in every thread -- 4 M uncached loads per launch -- measured 125 us
        // per launch: five times the finalize launches this replaces.)
        dr_acquire_agent();
    }
    __syncthreads();
}

// Streaming kernels below keep kBnRows independent 16-byte loads per thread and stream in flight: hipcc does
// not batch the loads of a "#pragma unroll"-ed grid-stride loop by itself (it waited for each one), which held
// these passes at ~3 TB/s of the ~8 TB/s HBM.
#ifndef DR_BN_ROWS
#define DR_BN_ROWS 4                     // experiment switch of the build (profiles/r02_experiments.md)
#endif
constexpr int kBnRows = DR_BN_ROWS;
// ... and twice as many where the raw output is stored as bf16 (the R16 variants): its loads are 8 bytes, the same bytes in flight
// (measured, visit 15: S=2 F=128 bf16 5554 -> 5633 crops/s, config 5 508 -> 518; 8 rows on the fp32 kernels: 2654 -> 2616)
#ifndef DR_BN_ROWS_R16
#define DR_BN_ROWS_R16 8
#endif
template <int R16> struct BnRows { static constexpr int value = R16 ? DR_BN_ROWS_R16 : kBnRows; };

// four consecutive channels of the raw conv output at element offset `e` (a multiple of 4): fp32 storage, or (R16) bf16 storage on
// the bf16 path -- half the bytes of the tensor every BatchReNorm pass reads.  A COMPILE-TIME variant of the streaming kernels:
// as a run-time branch around the loads it kept hipcc from issuing a thread's loads as one batch (measured: the backward reduce
// pass 4.8 -> 7.9 ms per three windows with HALF the raw bytes; profiles/r04_experiments.md section 9).
// DR_BN_NT (experiment switch of the build): bit 1 = the passes' 16-byte stores non-temporal, bit 2 = their 16-byte fp32 loads
#ifndef DR_BN_NT
#define DR_BN_NT 15                       // (measured: fp32 passes, visit 24: stores +0.25 %, loads +0.5 %, both +0.9-1.4 % on the step; the bf16 path's 8-byte accesses, visit 28: S=2 bf16 +0.4-0.9 %, config 5 equal)
#endif
__device__ __forceinline__ float4 bn_ld4(const float* p) {
#if (DR_BN_NT & 2) && !defined(DR_EMU)
    const dr_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const dr_f32x4*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
#else
    return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ void bn_st4(float* p, const float4 v) {
#if (DR_BN_NT & 1) && !defined(DR_EMU)
    const dr_f32x4 f = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(f, reinterpret_cast<dr_f32x4*>(p));
#else
    *reinterpret_cast<float4*>(p) = v;
#endif
}
// ... and (bits 4 / 8) the 8-byte bf16x4 stores / loads of the bf16 path's passes
__device__ __forceinline__ dr_bf16x4 bn_ld4h(const __bf16* p) {
#if (DR_BN_NT & 8) && !defined(DR_EMU)
    return __builtin_nontemporal_load(reinterpret_cast<const dr_bf16x4*>(p));
#else
    return *reinterpret_cast<const dr_bf16x4*>(p);
#endif
}
__device__ __forceinline__ void bn_st4h(__bf16* p, const dr_bf16x4 v) {
#if (DR_BN_NT & 4) && !defined(DR_EMU)
    __builtin_nontemporal_store(v, reinterpret_cast<dr_bf16x4*>(p));
#else
    *reinterpret_cast<dr_bf16x4*>(p) = v;
#endif
}
template <int R16>
__device__ __forceinline__ float4 bn_load_raw4(const float* raw, long e) {
    if (R16) {
        const dr_bf16x4 h = bn_ld4h(reinterpret_cast<const __bf16*>(raw) + e);
        const dr_f32x4 f = __builtin_convertvector(h, dr_f32x4);
        return make_float4(f[0], f[1], f[2], f[3]);
    }
    return bn_ld4(raw + e);
}
template <int R16>
__device__ __forceinline__ const float* bn_raw_advance(const float* raw, long elems) {
    return R16 ? reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(raw) + elems) : raw + elems;
}

// The per-channel inputs of bn_channel_coeffs, loaded BEFORE the partial rows are folded (unconditional: the shadow slots exist
// for every BatchReNorm layer): the finalize launches are 5 us chains of dependent round trips, this one now overlaps the fold's.
struct BnChanIn { float mm, mv, g, beta, sh_m, sh_v; };
__device__ __forceinline__ BnChanIn bn_channel_load(const BnTrainParams& p, int c) {
    BnChanIn in;
    in.mm = p.mm[c]; in.mv = p.mv[c]; in.g = p.gamma[c]; in.beta = p.beta[c];
    in.sh_m = p.shadow_mean[c]; in.sh_v = p.shadow_var[c];
    return in;
}
// The per-channel arithmetic of one (micro-)batch, ops.py:130-171, in three steps shared by every BatchReNorm forward path (the
// finalize launch, the self-folding apply pass, the look-back producers, the chain over micro-batch groups):
//   bn_channel_math     batch moments -> mean, inv_std, the clipped r / d against the moving statistics in `st`, scale | shift
//   bn_channel_store    the coefficients, to copy `gi` (micro-batch groups: fold_stride / bnc_stride floats apart; 0 otherwise)
//   bn_channel_advance  `st` -> what this batch leaves behind (moving averages, zero-debias accumulators; `step` = the batch's
//                       update count, for the debias correction);  bn_channel_store_state writes it back
struct BnChanOut { float scale, shift, mean, var, inv_std, r, d; };
__device__ __forceinline__ void bn_channel_moments(double sum, double sq, double cnt, float& mean, float& var) {
    const double mean_d = sum / cnt;
    double var_d = sq / cnt - mean_d * mean_d;
    if (var_d < 0.0) var_d = 0.0;
    mean = (float)mean_d; var = (float)var_d;
}
__device__ __forceinline__ BnChanOut bn_channel_math(const BnTrainParams& p, const BnChanIn& st, double sum, double sq, double cnt, float r_max,
                                                     float d_max) {
    BnChanOut o;
    bn_channel_moments(sum, sq, cnt, o.mean, o.var);
    const float std_b = sqrtf(o.var + p.eps);
    o.inv_std = 1.0f / std_b;
    const float mstd = sqrtf(st.mv + p.eps);
    float r = std_b / mstd;
    o.r = fminf(fmaxf(r, 1.0f / r_max), r_max);
    float d = (o.mean - st.mm) / mstd;
    o.d = fminf(fmaxf(d, -d_max), d_max);
    const float sc = o.inv_std * o.r;
    o.scale = sc * st.g;
    o.shift = (o.d - o.mean * sc) * st.g + st.beta;
    return o;
}
__device__ __forceinline__ void bn_channel_store(const BnTrainParams& p, int c, const BnChanOut& o, int gi) {
    float* scale = p.scale + (long)gi * p.fold_stride;
    float* shift = p.shift + (long)gi * p.fold_stride;
    float* bnc = p.bnc + (long)gi * p.bnc_stride;
    scale[c] = o.scale;
    shift[c] = o.shift;
    bnc[0 * p.C + c] = o.mean;
    bnc[1 * p.C + c] = o.inv_std;
    bnc[2 * p.C + c] = o.r;
    bnc[3 * p.C + c] = o.d;
}
// (the zero-debias correction of update `step`, ops.py:156-162; split off so that the chain over micro-batch groups can have it ready)
__device__ __forceinline__ float bn_debias_corr(const BnTrainParams& p, int step) { return 1.0f - powf(p.decay, (float)step); }
__device__ __forceinline__ void bn_channel_advance_with(const BnTrainParams& p, BnChanIn& st, float mean, float var, float corr) {
    const float om = 1.0f - p.decay;
    if (p.shadow_step > 0) {
        st.sh_m = st.sh_m - (st.sh_m - mean) * om;
        st.sh_v = st.sh_v - (st.sh_v - var) * om;
        st.mm = st.sh_m / corr;
        st.mv = st.sh_v / corr;
    } else {
        st.mm = st.mm - (st.mm - mean) * om;
        st.mv = st.mv - (st.mv - var) * om;
    }
}
__device__ __forceinline__ void bn_channel_advance(const BnTrainParams& p, BnChanIn& st, const BnChanOut& o, int step) {
    bn_channel_advance_with(p, st, o.mean, o.var, p.shadow_step > 0 ? bn_debias_corr(p, step) : 1.0f);
}
__device__ __forceinline__ void bn_channel_store_state(const BnTrainParams& p, int c, const BnChanIn& st) {
    if (p.shadow_step > 0) { p.shadow_mean[c] = st.sh_m; p.shadow_var[c] = st.sh_v; }
    p.mm_next[c] = st.mm;
    p.mv_next[c] = st.mv;
}

// one batch of M rows: coefficients out; `persist`: this caller also stores them and the updated state
__device__ __forceinline__ void bn_channel_coeffs(const BnTrainParams& p, int c, const BnChanIn& in, double sum, double sq, float& sc_out, float& sh_out,
                                                  bool persist) {
    const BnChanOut o = bn_channel_math(p, in, sum, sq, (double)p.M, p.r_max, p.d_max);
    sc_out = o.scale;
    sh_out = o.shift;
    if (persist) {
        BnChanIn st = in;
        bn_channel_store(p, c, o, 0);
        bn_channel_advance(p, st, o, p.shadow_step);
        bn_channel_store_state(p, c, st);
    }
}

// Fold the per-workgroup partials of the conv epilogue / the backward reduce and derive everything the step
// needs per channel.  Partials are stored [2][C][rows] -- the rows of one channel are contiguous -- so one wave folds
// one channel with coalesced loads and a fixed shuffle tree (reproducible); block = 4 waves = 4 channels.
// rows [row0, row0 + nrows) of the `rows_total` partial rows of channel c (a micro-batch group's share; the whole range otherwise)
__device__ __forceinline__ void fold_partials_wave(const double* part, int rows_total, int C, int c, int row0, int nrows, double& sum, double& sq);
__device__ __forceinline__ void fold_partials_wave(const double* part, int rows, int C, int c, double& sum, double& sq) {
    fold_partials_wave(part, rows, C, c, 0, rows, sum, sq);
}
__device__ __forceinline__ void fold_partials_wave(const double* part, int rows_total, int C, int c, int row0, int rows, double& sum, double& sq) {
    const int lane = threadIdx.x & 63;
    const double* pa = part + (long)c * rows_total + row0;
    const double* pb = part + ((long)C + c) * rows_total + row0;
    double a = 0.0, b = 0.0;
    // eight rows per lane and operand in flight, unconditional (clamped row; the value of a row past the end is dropped): a tail
    // loop of single loads was a round trip per 64 rows -- with the 640 rows of a 32x32 layer four trips where one does
    for (int r0 = lane; r0 < rows; r0 += 8 * 64) {
        double va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + u * 64, rc = r < rows ? r : rows - 1;
            va[u] = pa[rc]; vb[u] = pb[rc];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool ok = r0 + u * 64 < rows;
            a += ok ? va[u] : 0.0; b += ok ? vb[u] : 0.0;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
    }
    sum = a; sq = b;
}

// plain fold into out[0..C) = sum, out[C..2C) = sum of squares (test hook dr_dbg_conv2d)
__global__ __launch_bounds__(256) void stat_fold_kernel(const double* part, int rows, int C, double* out) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double sum, sq;
    fold_partials_wave(part, rows, C, c, sum, sq);
    if ((threadIdx.x & 63) == 0) { out[c] = sum; out[C + c] = sq; }
}

// dst[c] += sum of the first block of partial rows ([C][rows] doubles): the bias gradient of a conv whose column sums were
// produced by its reader's dgrad epilogue (conv_igemm.h, bst_act)
__global__ __launch_bounds__(256) void bias_grad_from_rows_kernel(const double* part, int rows, int C, float* dst) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double sum, unused;
    fold_partials_wave(part, rows, C, c, sum, unused);
    if ((threadIdx.x & 63) == 0) dst[c] += (float)sum;
}

// The finalize launches (forward here, backward below): ONE workgroup per channel, groups x fin_split waves.  Wave (g, sub) folds
// the sub-th share of micro-batch group g's partial rows (one coalesced round trip per 512 rows, all groups' loads in flight
// together), the shares meet in LDS, thread 0 adds them in a fixed order and walks the groups' chain of state updates in registers.
// (Rounds 1-3: four channels per workgroup, one wave per channel folding group after group -- a round trip per group behind
// the previous group's arithmetic, 14-23 us per launch at five groups and up to 0.4 ms where a 128x128 layer's 12 800 rows went
// through 32 workgroups; 134 such launches in each direction per window.)
constexpr int kBnFinalizeWaves = 16;
// partial rows one wave folds before its group gets a second wave (DR_BN_FIN_ROWS; test hook: dr_dbg_bn_finalize_rows)
inline int g_bn_finalize_rows = [] { const char* e = getenv("DR_BN_FIN_ROWS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();
inline int bn_finalize_split(int groups, int rows_per_group) {
    const int g = groups > 1 ? groups : 1;
    const int want = (rows_per_group + g_bn_finalize_rows - 1) / g_bn_finalize_rows, cap = kBnFinalizeWaves / g;
    return want < 1 ? 1 : (want > cap ? (cap < 1 ? 1 : cap) : want);
}
// this wave's share of its group's rows -> s_a / s_b[wave]; returns after the barrier
__device__ __forceinline__ void bn_finalize_fold(const double* part, int part_rows, int C, int c, int groups, int rows_per_group, int split,
                                                 double* s_a, double* s_b) {
    const int wave = threadIdx.x >> 6, g = wave / split, sub = wave - g * split;
    const int chunk = (rows_per_group + split - 1) / split;
    const int r0 = sub * chunk;
    int n = rows_per_group - r0;
    n = n > chunk ? chunk : (n < 0 ? 0 : n);
    double a = 0.0, b = 0.0;
    if (g < groups) fold_partials_wave(part, part_rows, C, c, g * rows_per_group + r0, n, a, b);
    if ((threadIdx.x & 63) == 0) { s_a[wave] = a; s_b[wave] = b; }
    __syncthreads();
}

__global__ __launch_bounds__(64 * kBnFinalizeWaves) void bn_fwd_finalize_kernel(const BnTrainParams p) {
    DR_PIN_ARGS(p.part, p.part_rows, p.C, p.M, p.beta, p.gamma, p.mm, p.mv, p.mm_next, p.mv_next, p.shadow_mean, p.shadow_var, p.shadow_step, p.scale, p.shift, p.bnc);
    __shared__ double s_a[kBnFinalizeWaves], s_b[kBnFinalizeWaves];
    const int c = blockIdx.x;
    const int G = p.groups > 1 ? p.groups : 1;
    const int rpg = p.groups > 1 ? p.rows_per_group : p.part_rows;
    BnChanIn st{};
    if (threadIdx.x < 64) st = bn_channel_load(p, c);         // beside the fold's round trip, not behind it (wave 0, every lane)
    bn_finalize_fold(p.part, p.part_rows, p.C, c, G, rpg, p.fin_split, s_a, s_b);
    if (threadIdx.x >= 64) return;
    // wave 0.  Lane g: group g.  What does not depend on the chain -- the batch moments, the debias correction (a powf) -- in
    // parallel; the chain itself (moving statistics after group g-1 -> before group g: two multiply-adds and two divisions per
    // group, ops.py:156-162) in every lane; then lane g finishes its group against the state it caught on the way.  Same
    // arithmetic per group as bn_channel_math / bn_channel_advance (one thread, group after group: 7 us of dependent
    // arithmetic at five groups).
    const int lane = threadIdx.x;
    const double cnt = p.groups > 1 ? (double)p.Mg : (double)p.M;
    double sum = 0.0, sq = 0.0;
    float mean = 0.f, var = 0.f, corr = 1.f, r_max = p.r_max, d_max = p.d_max;
    if (lane < G) {
        for (int k = 0; k < p.fin_split; ++k) { sum += s_a[lane * p.fin_split + k]; sq += s_b[lane * p.fin_split + k]; }
        bn_channel_moments(sum, sq, cnt, mean, var);
        if (p.shadow_step > 0) corr = bn_debias_corr(p, p.shadow_step + lane);
    }
    if (p.groups > 1) {
#pragma unroll
        for (int g = 0; g < kMaxGroups; ++g)
            if (lane == g) { r_max = p.r_max_g[g]; d_max = p.d_max_g[g]; }
    }
    BnChanIn cur = st, mine = st;
    for (int g = 0; g < G; ++g) {
        const float m_g = __shfl(mean, g), v_g = __shfl(var, g), c_g = __shfl(corr, g);
        if (lane == g) mine = cur;                            // the state group g reads: what group g-1 left
        bn_channel_advance_with(p, cur, m_g, v_g, c_g);
    }
    if (lane < G) bn_channel_store(p, c, bn_channel_math(p, mine, sum, sq, cnt, r_max, d_max), lane);
    if (lane == 0) bn_channel_store_state(p, c, cur);
}
inline void launch_bn_fwd_finalize(BnTrainParams& p, hipStream_t s) {
    const int G = p.groups > 1 ? p.groups : 1;
    p.fin_split = bn_finalize_split(G, p.groups > 1 ? p.rows_per_group : p.part_rows);
    DR_LAUNCH(bn_fwd_finalize_kernel, dim3(p.C), dim3(64 * G * p.fin_split), 0, s, p);
}

// FUSE: layers with few partial rows (everything at 8x8 and below) skip the finalize launch -- every workgroup
// folds the rows itself (serially per channel, fixed order, a few L2 hits) and workgroup 0 persists the results.
// MODE 0: coefficients come from a bn_fwd_finalize_kernel launch; 1 (FUSE): few partial rows, every workgroup folds them;
// 2: look-back hand-off (above).
template <int MODE, int R16 = 0>
__global__ __launch_bounds__(256) void bn_train_apply_kernel(const BnTrainParams p_in) {
    constexpr int kRows = BnRows<R16>::value;                 // rows (independent loads per tensor) a thread keeps in flight
    DR_PIN_ARGS(p_in.raw, p_in.raw_cs, p_in.M, p_in.C, p_in.scale, p_in.shift, p_in.relu, p_in.res.p, p_in.res.cs, p_in.res.coff, p_in.out.p, p_in.out.cs, p_in.out.coff, p_in.out_bf16, p_in.part, p_in.part_rows, (int)gridDim.x);
    BnTrainParams p = p_in;
    if (MODE == 0 && p.groups > 1) {                      // blockIdx.y = micro-batch group: its rows, its coefficients
        const long g = blockIdx.y, r0 = g * p.Mg;
        p.raw = bn_raw_advance<R16>(p.raw, r0 * p.raw_cs);
        if (p.res.p) p.res.p += r0 * p.res.cs;
        if (p.out_bf16) p.out.p = reinterpret_cast<float*>(reinterpret_cast<__bf16*>(p.out.p) + r0 * p.out.cs);
        else p.out.p += r0 * p.out.cs;
        p.scale += g * p.fold_stride; p.shift += g * p.fold_stride;
        p.M = p.Mg;
    }
    constexpr bool FUSE = MODE == 1;
    __shared__ float s_sc[FUSE ? 1024 : 1], s_sh[FUSE ? 1024 : 1];
    // MODE 2: the grid is [ceil(C/4) producer workgroups | the streaming workgroups].  A producer folds its four channels,
    // publishes and EXITS -- it never waits, so producers cannot starve each other however few workgroups run at a time.
    const int nprod = MODE == 2 ? (p.C + 3) >> 2 : 0;
    if (MODE == 2) {
        if ((int)blockIdx.x < nprod) {
            const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
            if (c < p.C) {
                const BnChanIn in = bn_channel_load(p, c);
                double sum, sq;
                fold_partials_wave(p.part, p.part_rows, p.C, c, sum, sq);
                if ((threadIdx.x & 63) == 0) {
                    float sc, sh;
                    bn_channel_coeffs(p, c, in, sum, sq, sc, sh, true);
                }
            }
            bn_handoff_publish(p.flag, 1);
            return;
        }
        bn_handoff_wait(p.flag, p.flag_target);
    }
    const int bid = (int)blockIdx.x - nprod, nblk = (int)gridDim.x - nprod;      // this workgroup among the streaming ones
    if (FUSE) {
        for (int c = threadIdx.x; c < p.raw_cs; c += 256) {
            float sc = 0.f, sh = 0.f;
            if (c < p.C) {
                const BnChanIn in = bn_channel_load(p, c);
                double sum = 0.0, sq = 0.0;
                for (int r = 0; r < p.part_rows; ++r) {
                    sum += p.part[(long)c * p.part_rows + r];
                    sq += p.part[((long)p.C + c) * p.part_rows + r];
                }
                bn_channel_coeffs(p, c, in, sum, sq, sc, sh, bid == 0);
            }
            s_sc[c] = sc; s_sh[c] = sh;
        }
        __syncthreads();
    }
    const int c4n = p.raw_cs / 4;                  // channel groups per row (<= 256)
    const int rpb = 256 / c4n;                     // rows per workgroup pass
    const int cg = threadIdx.x % c4n, rp = threadIdx.x / c4n;
    if (rp >= rpb) return;
    float sc[4], sh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = cg * 4 + k;
        if (FUSE) { sc[k] = s_sc[c]; sh[k] = s_sh[c]; }
        else {
            // UNCONDITIONAL loads (index clamped; a pad channel computes garbage that no store keeps): behind "c < C ? load : 0"
            // hipcc parks an s_waitcnt vmcnt(0) at the join, and the coefficient round trip (written by the finalize launch on
            // other XCDs: an L2 miss) was paid before the first streaming load was even issued -- two dependent round trips in
            // launches that last 5 us.
            const int cc = c < p.C ? c : p.C - 1;
            sc[k] = p.scale[cc];                                // written by bn_fwd_finalize_kernel
            sh[k] = p.shift[cc];
        }
    }
    const bool full = cg * 4 + 4 <= p.C;
    const bool vec_out = full && (p.out.coff % 4 == 0) && (p.out.cs % 4 == 0);
    const bool vec_res = p.res.p && full && (p.res.coff % 4 == 0) && (p.res.cs % 4 == 0);
    const long stride = (long)nblk * rpb;
    for (long m0 = (long)bid * rpb + rp; m0 < p.M; m0 += stride * kRows) {
        float4 x[kRows], rv[kRows];
#pragma unroll
        for (int u = 0; u < kRows; ++u) {
            const long m = m0 + u * stride, mc = m < p.M ? m : p.M - 1;     // tail rows re-read the last row
            x[u] = bn_load_raw4<R16>(p.raw, mc * p.raw_cs + cg * 4);
        }
        if (vec_res) {
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                const long m = m0 + u * stride, mc = m < p.M ? m : p.M - 1;
                rv[u] = bn_ld4(p.res.p + mc * p.res.cs + p.res.coff + cg * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < kRows; ++u) {
            const long m = m0 + u * stride;
            if (m >= p.M) break;
            float v[4] = {x[u].x * sc[0] + sh[0], x[u].y * sc[1] + sh[1], x[u].z * sc[2] + sh[2], x[u].w * sc[3] + sh[3]};
            if (p.relu) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
            }
            if (vec_res) {
                v[0] += rv[u].x; v[1] += rv[u].y; v[2] += rv[u].z; v[3] += rv[u].w;
            } else if (p.res.p) {
                const float* rs = p.res.p + m * p.res.cs + p.res.coff + cg * 4;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (cg * 4 + k < p.C) v[k] += rs[k];
            }
            float* o = p.out.p + m * p.out.cs + p.out.coff + cg * 4;
            if (p.out_bf16) {                                               // pad channels of the group: zero, like every producer of bf16 storage
                const dr_f32x4 f = {cg * 4 + 0 < p.C ? v[0] : 0.f, cg * 4 + 1 < p.C ? v[1] : 0.f, cg * 4 + 2 < p.C ? v[2] : 0.f, cg * 4 + 3 < p.C ? v[3] : 0.f};
                bn_st4h(reinterpret_cast<__bf16*>(p.out.p) + m * p.out.cs + p.out.coff + cg * 4, __builtin_convertvector(f, dr_bf16x4));
            } else if (vec_out) {
                bn_st4(o, make_float4(v[0], v[1], v[2], v[3]));
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (cg * 4 + k < p.C) o[k] = v[k];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// BatchReNorm backward.
//   out = gamma*(r*yhat + d) + beta, yhat = (x-mean)*inv_std, g = dOut * [out_pre_relu > 0]
//   dbeta = sum g ; dgamma = r*sum(g*yhat) + d*sum(g)
//   dx = gamma*r*inv_std * (g - mean(g) - yhat*mean(g*yhat))
// reduce: per-channel sum g, sum g*yhat -- one fp64 partial row per workgroup (no floating-point atomics);
// finalize: folds the rows in a fixed order, writes the three dx coefficients, accumulates dbeta/dgamma;
// apply: dx from the coefficients.
// ------------------------------------------------------------------------------------------------
struct BnBwdParams {
    View dout; const float* raw; int raw_cs; long M; int C; int relu;
    const float* scale; const float* shift; const float* bnc; const float* gamma;
    double* part; int part_rows;      // [2][C][part_rows]: per-workgroup sum g / sum g*yhat of the reduce pass
    float* coef;                      // [3][C]: c1 = gamma*r*inv_std, c2 = mean(g), c3 = mean(g*yhat) (finalize -> apply)
    float* dbeta; float* dgamma;      // flat-gradient slices (accumulated)
    float* draw;                      // out: gradient wrt the raw conv output, dense stride raw_cs
    View dres; int dres_acc;          // apply pass, nullable: residual source's gradient (+)= dOut (out = act(..) + res)
    int* flag; int flag_target;       // look-back hand-off (bn_bwd_apply_kernel<2>)
    int raw_bf16;                     // raw holds bf16 elements (BnTrainParams::raw_bf16)
    int dout_bf16;                    // dout holds bf16 elements (stride dout.cs elements; bn_bwd_apply_kernel's D16 variant)
    int draw_bf16;                    // draw holds bf16 elements (same element stride raw_cs): both of its readers -- the layer's
                                      // dgrad and weight gradient on the bf16 matrix cores -- round it to bf16 anyway
                                      // while staging, so the numbers are the same and the tensor is half the bytes
    // micro-batch groups (BnTrainParams): per-group sums, coefficients (coef of group g: 3*C floats behind group g-1's) and
    // forward values (scale | shift, bnc copies fold_stride / bnc_stride floats apart); dbeta / dgamma sum over the groups
    int groups; int rows_per_group; long Mg; long fold_stride; long bnc_stride;
    int fin_split;                              // finalize launch: waves per group (launch_bn_*_finalize sets it)
};

template <int R16 = 0>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const BnBwdParams p_in) {
    constexpr int kRows = BnRows<R16>::value;                 // rows (independent loads per tensor) a thread keeps in flight
    DR_PIN_ARGS(p_in.dout.p, p_in.dout.cs, p_in.dout.coff, p_in.raw, p_in.raw_cs, p_in.M, p_in.C, p_in.relu, p_in.scale, p_in.shift, p_in.bnc, p_in.part, (int)gridDim.x);
    BnBwdParams p = p_in;
    int part_row = (int)blockIdx.x, part_rows = (int)gridDim.x;
    if (p.groups > 1) {                                   // blockIdx.y = micro-batch group
        const long g = blockIdx.y, r0 = g * p.Mg;
        p.dout.p += r0 * p.dout.cs;
        p.raw = bn_raw_advance<R16>(p.raw, r0 * p.raw_cs);
        p.scale += g * p.fold_stride; p.shift += g * p.fold_stride; p.bnc += g * p.bnc_stride;
        p.M = p.Mg;
        part_row += (int)g * (int)gridDim.x; part_rows *= (int)gridDim.y;
    }
    __shared__ double s1[256 * 4];
    __shared__ double s2[256 * 4];
    const int c4n = p.raw_cs / 4;
    const int rpb = 256 / c4n;
    const int tid = threadIdx.x;
    const int cg = tid % c4n, rp = tid / c4n;
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    if (rp < rpb) {
        float sc[4], sh[4], mean[4], istd[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = cg * 4 + k, cc = c < p.C ? c : p.C - 1;          // unconditional loads, see bn_train_apply_kernel
            sc[k] = p.scale[cc]; sh[k] = p.shift[cc];
            mean[k] = p.bnc[cc]; istd[k] = p.bnc[p.C + cc];
        }
        const bool full = cg * 4 + 4 <= p.C;
        const bool vec_d = full && (p.dout.coff % 4 == 0) && (p.dout.cs % 4 == 0);
        const long stride = (long)gridDim.x * rpb;
        for (long m0 = (long)blockIdx.x * rpb + rp; m0 < p.M; m0 += stride * kRows) {
            float4 x4[kRows], d4[kRows];
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                const long m = m0 + u * stride, mc = m < p.M ? m : p.M - 1;
                x4[u] = bn_load_raw4<R16>(p.raw, mc * p.raw_cs + cg * 4);
            }
            if (vec_d) {
#pragma unroll
                for (int u = 0; u < kRows; ++u) {
                    const long m = m0 + u * stride, mc = m < p.M ? m : p.M - 1;
                    d4[u] = bn_ld4(p.dout.p + mc * p.dout.cs + p.dout.coff + cg * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                const long m = m0 + u * stride;
                if (m >= p.M) break;
                const float x[4] = {x4[u].x, x4[u].y, x4[u].z, x4[u].w};
                float g[4] = {0.f, 0.f, 0.f, 0.f};
                if (vec_d) {
                    g[0] = d4[u].x; g[1] = d4[u].y; g[2] = d4[u].z; g[3] = d4[u].w;
                } else {
                    const float* dp = p.dout.p + m * p.dout.cs + p.dout.coff + cg * 4;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (cg * 4 + k < p.C) g[k] = dp[k];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (p.relu && !(x[k] * sc[k] + sh[k] > 0.f)) g[k] = 0.f;
                    const float yh = (x[k] - mean[k]) * istd[k];
                    a[k] += (double)g[k];
                    b[k] += (double)g[k] * (double)yh;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { s1[tid * 4 + k] = a[k]; s2[tid * 4 + k] = b[k]; }
    __syncthreads();
    if (tid < c4n) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = tid * 4 + k;
            if (c < p.C) {
                double ta = 0.0, tb = 0.0;
                for (int r = 0; r < rpb; ++r) { ta += s1[(r * c4n + tid) * 4 + k]; tb += s2[(r * c4n + tid) * 4 + k]; }
                p.part[(long)c * part_rows + part_row] = ta;                       // [2][C][rows]
                p.part[((long)p.C + c) * part_rows + part_row] = tb;
            }
        }
    }
}

// (mapping: bn_fwd_finalize_kernel)  coefficients per group; dbeta / dgamma summed over the groups in order
__global__ __launch_bounds__(64 * kBnFinalizeWaves) void bn_bwd_finalize_kernel(const BnBwdParams p) {
    DR_PIN_ARGS(p.part, p.part_rows, p.C, p.M, p.gamma, p.bnc, p.coef, p.dbeta, p.dgamma);
    __shared__ double s_a[kBnFinalizeWaves], s_b[kBnFinalizeWaves];
    const int c = blockIdx.x;
    const int G = p.groups > 1 ? p.groups : 1;
    const int rpg = p.groups > 1 ? p.rows_per_group : p.part_rows;
    const double Mg = p.groups > 1 ? (double)p.Mg : (double)p.M;
    // wave 0, lane g: group g.  Everything the last lines read, before the fold (one round trip beside the fold's, not behind it)
    float r = 0.f, d = 0.f, istd = 0.f, gam = 0.f, db = 0.f, dg = 0.f;
    if (threadIdx.x < 64) {
        const float* bnc = p.bnc + (long)((int)threadIdx.x < G ? (int)threadIdx.x : 0) * p.bnc_stride;
        r = bnc[2 * p.C + c]; d = bnc[3 * p.C + c]; istd = bnc[p.C + c];
        gam = p.gamma[c]; db = p.dbeta[c]; dg = p.dgamma[c];
    }
    bn_finalize_fold(p.part, p.part_rows, p.C, c, G, rpg, p.fin_split, s_a, s_b);
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    double sg = 0.0, sgy = 0.0;
    if (lane < G) {
        for (int k = 0; k < p.fin_split; ++k) { sg += s_a[lane * p.fin_split + k]; sgy += s_b[lane * p.fin_split + k]; }
        float* coef = p.coef + (long)lane * 3 * p.C;
        coef[0 * p.C + c] = gam * r * istd;
        coef[1 * p.C + c] = (float)(sg / Mg);
        coef[2 * p.C + c] = (float)(sgy / Mg);
    }
    const float fsg = (float)sg, fsgy = (float)sgy;
    for (int g = 0; g < G; ++g) {                             // dbeta / dgamma: the groups' terms added in order
        const float a = __shfl(fsg, g), b = __shfl(fsgy, g), rg = __shfl(r, g), dgp = __shfl(d, g);
        db = db + a;
        dg = dg + (rg * b + dgp * a);
    }
    if (lane == 0) { p.dbeta[c] = db; p.dgamma[c] = dg; }
}
inline void launch_bn_bwd_finalize(BnBwdParams& p, hipStream_t s) {
    const int G = p.groups > 1 ? p.groups : 1;
    p.fin_split = bn_finalize_split(G, p.groups > 1 ? p.rows_per_group : p.part_rows);
    DR_LAUNCH(bn_bwd_finalize_kernel, dim3(p.C), dim3(64 * G * p.fin_split), 0, s, p);
}

// MODE 0: coefficients from a bn_bwd_finalize_kernel launch; 1 (FUSE): fold the few partial rows here; 2: look-back hand-off
// D16 (with R16, the bf16 path): dOut holds bf16 elements too (BnBwdParams::dout_bf16) -- written that way by the epilogue of the
// tensor's only reader's input-gradient launch; whole channel groups of four (the executor stores a gradient as bf16 only then)
template <int MODE, int R16 = 0, int D16 = 0>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnBwdParams p_in) {
    constexpr int kRows = BnRows<R16>::value;                 // rows (independent loads per tensor) a thread keeps in flight
    DR_PIN_ARGS(p_in.dout.p, p_in.dout.cs, p_in.dout.coff, p_in.raw, p_in.raw_cs, p_in.M, p_in.C, p_in.relu, p_in.scale, p_in.shift, p_in.bnc, p_in.coef, p_in.draw, p_in.dres.p, p_in.dres.cs, p_in.dres.coff, p_in.dres_acc, p_in.draw_bf16, p_in.part, p_in.part_rows);
    BnBwdParams p = p_in;
    if (MODE == 0 && p.groups > 1) {                      // blockIdx.y = micro-batch group: its rows, its coefficients
        const long g = blockIdx.y, r0 = g * p.Mg;
        p.dout.p = D16 ? reinterpret_cast<float*>(reinterpret_cast<__bf16*>(p.dout.p) + r0 * p.dout.cs) : p.dout.p + r0 * p.dout.cs;
        p.raw = bn_raw_advance<R16>(p.raw, r0 * p.raw_cs);
        if (p.draw_bf16) p.draw = reinterpret_cast<float*>(reinterpret_cast<__bf16*>(p.draw) + r0 * p.raw_cs);
        else p.draw += r0 * p.raw_cs;
        if (p.dres.p) p.dres.p += r0 * p.dres.cs;
        p.scale += g * p.fold_stride; p.shift += g * p.fold_stride; p.bnc += g * p.bnc_stride; p.coef += g * 3 * p.C;
        p.M = p.Mg;
    }
    constexpr bool FUSE = MODE == 1;
    __shared__ float s_c[FUSE ? 3 : 1][FUSE ? 1024 : 1];
    const int nprod = MODE == 2 ? (p.C + 3) >> 2 : 0;          // [producers | streaming workgroups], see bn_train_apply_kernel
    if (MODE == 2) {
        if ((int)blockIdx.x < nprod) {
            const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
            if (c < p.C) {
                double sg, sgy;
                fold_partials_wave(p.part, p.part_rows, p.C, c, sg, sgy);
                if ((threadIdx.x & 63) == 0) {
                    const float r = p.bnc[2 * p.C + c], d = p.bnc[3 * p.C + c], istd = p.bnc[p.C + c];
                    p.coef[0 * p.C + c] = p.gamma[c] * r * istd;
                    p.coef[1 * p.C + c] = (float)(sg / (double)p.M);
                    p.coef[2 * p.C + c] = (float)(sgy / (double)p.M);
                    p.dbeta[c] += (float)sg;
                    p.dgamma[c] += r * (float)sgy + d * (float)sg;
                }
            }
            bn_handoff_publish(p.flag, 1);
            return;
        }
        bn_handoff_wait(p.flag, p.flag_target);
    }
    const int bid = (int)blockIdx.x - nprod, nblk = (int)gridDim.x - nprod;
    if (FUSE) {
        for (int c = threadIdx.x; c < p.raw_cs; c += 256) {
            float k1 = 0.f, k2 = 0.f, k3 = 0.f;
            if (c < p.C) {
                double sg = 0.0, sgy = 0.0;
                for (int r = 0; r < p.part_rows; ++r) {
                    sg += p.part[(long)c * p.part_rows + r];
                    sgy += p.part[((long)p.C + c) * p.part_rows + r];
                }
                const float r_ = p.bnc[2 * p.C + c], d_ = p.bnc[3 * p.C + c], istd_ = p.bnc[p.C + c];
                k1 = p.gamma[c] * r_ * istd_;
                k2 = (float)(sg / (double)p.M);
                k3 = (float)(sgy / (double)p.M);
                if (bid == 0) {
                    p.dbeta[c] += (float)sg;
                    p.dgamma[c] += r_ * (float)sgy + d_ * (float)sg;
                }
            }
            s_c[0][c] = k1; s_c[1][c] = k2; s_c[2][c] = k3;
        }
        __syncthreads();
    }
    const int c4n = p.raw_cs / 4;
    const int rpb = 256 / c4n;
    const int cg = threadIdx.x % c4n, rp = threadIdx.x / c4n;
    if (rp >= rpb) return;
    float sc[4], sh[4], mean[4], istd[4], c1[4], c2[4], c3[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = cg * 4 + k, cc = c < p.C ? c : p.C - 1;              // unconditional loads, see bn_train_apply_kernel
        sc[k] = p.scale[cc]; sh[k] = p.shift[cc]; mean[k] = p.bnc[cc]; istd[k] = p.bnc[p.C + cc];
        if (FUSE) { c1[k] = s_c[0][cc]; c2[k] = s_c[1][cc]; c3[k] = s_c[2][cc]; }
        else { c1[k] = p.coef[cc]; c2[k] = p.coef[p.C + cc]; c3[k] = p.coef[2 * p.C + cc]; }
    }
    const bool full = cg * 4 + 4 <= p.C;
    const bool vec_d = full && (p.dout.coff % 4 == 0) && (p.dout.cs % 4 == 0);
    const bool vec_r = full && p.dres.p && (p.dres.coff % 4 == 0) && (p.dres.cs % 4 == 0);
    const long stride = (long)nblk * rpb;
    for (long m0 = (long)bid * rpb + rp; m0 < p.M; m0 += stride * kRows) {
        float4 x4[kRows], d4[kRows], r4[kRows];
#pragma unroll
        for (int u = 0; u < kRows; ++u) {
            const long m = m0 + u * stride, mc = m < p.M ? m : p.M - 1;
            x4[u] = bn_load_raw4<R16>(p.raw, mc * p.raw_cs + cg * 4);
        }
        if (D16) {
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                const long m = m0 + u * stride, mc = m < p.M ? m : p.M - 1;
                const dr_bf16x4 hd = bn_ld4h(reinterpret_cast<const __bf16*>(p.dout.p) + mc * p.dout.cs + p.dout.coff + cg * 4);
                const dr_f32x4 fd = __builtin_convertvector(hd, dr_f32x4);
                d4[u] = make_float4(fd[0], fd[1], fd[2], fd[3]);
            }
        } else if (vec_d) {
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                const long m = m0 + u * stride, mc = m < p.M ? m : p.M - 1;
                d4[u] = bn_ld4(p.dout.p + mc * p.dout.cs + p.dout.coff + cg * 4);
            }
        }
        const bool acc_r = vec_r && p.dres_acc;                       // the residual gradient this pass adds to: same batch of loads
        if (acc_r) {
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                const long m = m0 + u * stride, mc = m < p.M ? m : p.M - 1;
                r4[u] = bn_ld4(p.dres.p + mc * p.dres.cs + p.dres.coff + cg * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < kRows; ++u) {
            const long m = m0 + u * stride;
            if (m >= p.M) break;
            const float x[4] = {x4[u].x, x4[u].y, x4[u].z, x4[u].w};
            float g[4] = {0.f, 0.f, 0.f, 0.f};
            if (D16 || vec_d) {
                g[0] = d4[u].x; g[1] = d4[u].y; g[2] = d4[u].z; g[3] = d4[u].w;
            } else {
                const float* dp = p.dout.p + m * p.dout.cs + p.dout.coff + cg * 4;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (cg * 4 + k < p.C) g[k] = dp[k];
            }
            if (p.dres.p) {                                           // d(res) = dOut, unmasked: the add comes after the ReLU
                float* q = p.dres.p + m * p.dres.cs + p.dres.coff + cg * 4;
                if (vec_r) {
                    float4 v = make_float4(g[0], g[1], g[2], g[3]);
                    if (acc_r) { v.x += r4[u].x; v.y += r4[u].y; v.z += r4[u].z; v.w += r4[u].w; }
                    bn_st4(q, v);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (cg * 4 + k < p.C) q[k] = p.dres_acc ? q[k] + g[k] : g[k];
                }
            }
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (p.relu && !(x[k] * sc[k] + sh[k] > 0.f)) g[k] = 0.f;
                const float yh = (x[k] - mean[k]) * istd[k];
                o[k] = (cg * 4 + k < p.C) ? c1[k] * (g[k] - c2[k] - yh * c3[k]) : 0.f;
            }
            if (p.draw_bf16) {
                const dr_f32x4 f = {o[0], o[1], o[2], o[3]};
                bn_st4h(reinterpret_cast<__bf16*>(p.draw) + m * p.raw_cs + cg * 4, __builtin_convertvector(f, dr_bf16x4));
            } else {
                bn_st4(p.draw + m * p.raw_cs + cg * 4, make_float4(o[0], o[1], o[2], o[3]));
            }
        }
    }
}

// bias convs: g(m,c) = dOut(m,c) * factor * [out(m,c) > 0 if relu]  -> dense scratch (stride gcs), pads zero
__global__ __launch_bounds__(256) void act_bwd_kernel(View dout, View out, int relu, float factor, float* g, int gcs, long M,
                                                      int C) {
    const int c4n = gcs / 4;
    const long total = M * c4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c0 = int(i % c4n) * 4;
        const long m = i / c4n;
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + k;
            o[k] = 0.f;
            if (c < C) {
                float v = dout.p[m * dout.cs + dout.coff + c] * factor;
                if (relu && !(out.p[m * out.cs + out.coff + c] > 0.f)) v = 0.f;
                o[k] = v;
            }
        }
        *reinterpret_cast<float4*>(g + m * gcs + c0) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// part[blockIdx.x][c] = sum over this workgroup's rows of g(m,c)  (bias gradient; wgrad_reduce_kernel folds the
// rows into the flat gradient in a fixed order -- no floating-point atomics)
__global__ __launch_bounds__(256) void colsum_kernel(const float* g, int cs, int coff, long M, int C, float* part) {
    __shared__ double s1[256];
    const int tid = threadIdx.x;
    const int cpb = C < 256 ? C : 256;
    const int rows_par = 256 / cpb;
    const long stride = (long)gridDim.x * rows_par;
    for (int c0 = 0; c0 < C; c0 += cpb) {
        const int c = c0 + tid % cpb;
        const int rp = tid / cpb;
        double a = 0.0;
        if (rp < rows_par && c < C) {
            long m = (long)blockIdx.x * rows_par + rp;
            const float* col = g + coff + c;
            for (; m + 7 * stride < M; m += 8 * stride) {          // 8 independent loads in flight
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = col[(m + u * stride) * cs];
#pragma unroll
                for (int u = 0; u < 8; ++u) a += (double)v[u];
            }
            for (; m < M; m += stride) a += (double)col[m * cs];
        }
        s1[tid] = a;
        __syncthreads();
        if (tid < cpb && c0 + tid < C) {
            double ta = 0.0;
            for (int r = 0; r < rows_par; ++r) ta += s1[r * cpb + tid];
            part[(long)blockIdx.x * C + c0 + tid] = (float)ta;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Max-pool backward, as a GATHER over the recorded arg-max (maxpool_kernel): thread = (input pixel, 4 channels); each of the
// up to 2 x 2 windows that contain the pixel hands its dOut over iff its first maximum (scan order ky, kx; strict >, the
// convention of torch's max_pool2d which the oracle's autograd follows) sits on this pixel.  Every input element is written
// exactly once, in a fixed order: no floating-point atomics, no clearing of dx (acc = 0: dx is overwritten).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const unsigned char* arg, float* dx, int dx_cs, int dx_coff, int B, int H, int W,
                                                          int C, int k, int pad_t, int pad_l, const float* dy, int y_cs,
                                                          int y_coff, int Ho, int Wo, int acc) {
    const int c4n = C / 4;
    const long total = (long)B * H * W * c4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = int(i % c4n);
        const long ip = i / c4n;
        const int ix = int(ip % W);
        const int iy = int((ip / W) % H);
        const int b = int(ip / ((long)W * H));
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        // windows (oy, ox) with oy*2 + ky - pad_t == iy for some ky in [0, k)
        const int oy_hi = (iy + pad_t) >> 1, ox_hi = (ix + pad_l) >> 1;
        for (int oy = oy_hi; oy >= 0 && oy * 2 + k - 1 - pad_t >= iy; --oy) {
            if (oy >= Ho) continue;
            const int ky = iy + pad_t - oy * 2;
            for (int ox = ox_hi; ox >= 0 && ox * 2 + k - 1 - pad_l >= ix; --ox) {
                if (ox >= Wo) continue;
                const int kx = ix + pad_l - ox * 2;
                const long op = ((long)b * Ho + oy) * Wo + ox;
                const unsigned a4 = *reinterpret_cast<const unsigned*>(arg + op * C + c4 * 4);
                const float4 d = *reinterpret_cast<const float4*>(dy + op * y_cs + y_coff + c4 * 4);
                const unsigned t = (unsigned)(ky * k + kx);
                if ((a4 & 0xFFu) == t) g.x += d.x;
                if (((a4 >> 8) & 0xFFu) == t) g.y += d.y;
                if (((a4 >> 16) & 0xFFu) == t) g.z += d.z;
                if ((a4 >> 24) == t) g.w += d.w;
            }
        }
        float4* q = reinterpret_cast<float4*>(dx + ip * dx_cs + dx_coff + c4 * 4);
        if (acc) { const float4 o = *q; g.x += o.x; g.y += o.y; g.z += o.z; g.w += o.w; }
        *q = g;
    }
}

// upsample-add backward: da (+)= dout ; dlo(y/2,x/2) (+)= sum of the 2x2 block of dout.  thread = (lo pixel, channel)
// acc_a / acc_lo = 0: this is the first gradient written into that buffer (overwrite, buffer not zeroed).
__global__ __launch_bounds__(256) void upsample_add_bwd_kernel(View dout, View da, View dlo, int B, int H, int W, int C,
                                                               int acc_a, int acc_lo) {
    const int h2 = H / 2, w2 = W / 2;
    const long total = (long)B * h2 * w2 * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = int(i % C);
        const long lp = i / C;
        const int lx = int(lp % w2);
        const int ly = int((lp / w2) % h2);
        const int b = int(lp / ((long)w2 * h2));
        float s = 0.f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const long pix = ((long)b * H + ly * 2 + dy) * W + lx * 2 + dx;
                const float g = dout.p[pix * dout.cs + dout.coff + c];
                float* q = da.p + pix * da.cs + da.coff + c;
                *q = acc_a ? (*q + g) : g;
                s += g;
            }
        float* q = dlo.p + lp * dlo.cs + dlo.coff + c;
        *q = acc_lo ? (*q + s) : s;
    }
}

// ------------------------------------------------------------------------------------------------
// Loss: target synthesis + l2 losses + gradient seeds for every stack, one thread per map pixel.
// fp32 in the reference's op order (contraction off), see oracle/pose.py make_targets.
// ------------------------------------------------------------------------------------------------
constexpr int kMaxStack = 8;
struct LossParams {
    View hm[kMaxStack], hm3[kMaxStack], um[kMaxStack];        // predictions
    View dhm[kMaxStack], dhm3[kMaxStack], dum[kMaxStack];     // gradient seeds (written)
    int S, B, h, w, J;
    const float* tiny; const float* pose; const float* cfg; const float* com;
    double* acc;                                              // [3] hm, hm3, um (sum x^2 / 2)
};

__global__ __launch_bounds__(256) void loss_kernel(const LossParams p) {
#pragma clang fp contract(off)
    // One thread per (pixel, joint), joints fastest: the threads of a wave read and write consecutive channels of consecutive
    // pixels (the maps are [pixel][J] / [pixel][3J]), every map element is touched by exactly one thread.  (One thread per pixel
    // walking the joints, one joint per workgroup row: a wave touched 64 cache lines for 64 floats, and the J rows of the grid read
    // every line J times -- 480 us for a 200-crop window, 230 MB of maps.)  The per-pixel point cloud is recomputed per joint: a
    // dozen flops.
    __shared__ double red[3][4];
    const int npix = p.h * p.w;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    double l_hm = 0.0, l_hm3 = 0.0, l_um = 0.0;
    if (idx < (long)p.B * npix * p.J) {
        const int j = (int)(idx % p.J);
        const int i = (int)(idx / p.J);
        const int b = i / npix, px = i % npix;
        const float* cfg = p.cfg + b * 6;
        const float cx0 = p.com[b * 3 + 0], cy0 = p.com[b * 3 + 1], cz0 = p.com[b * 3 + 2];
        const float w_ratio = cfg[4] / (float)p.w, h_ratio = cfg[5] / (float)p.h;
        const float fx = cfg[0] / w_ratio, fy = cfg[1] / h_ratio, cx = cfg[2] / w_ratio, cy = cfg[3] / h_ratio;
        const float xx = (float)(px % p.w), yy = (float)(px / p.w);
        // point cloud at this pixel (preprocess.py:202-225)
        const float t = p.tiny[i];
        const float min_depth = cz0 - 300.0f * 0.5f, max_depth = cz0 + 300.0f * 0.5f;
        const float zz = (t < -0.99f) ? max_depth : (t * 300.0f + min_depth);
        float X = (xx - cx) * (zz / fx);
        float Y = (yy - cy) * (zz / fy);
        X = (X - cx0) / 100.0f;
        Y = (Y - cy0) / 100.0f;
        const float Z = (zz - cz0) / 100.0f;
        const float* ps = p.pose + (long)b * 3 * p.J + 3 * j;
        // 2D cone (:225-242)
        const float u = ps[0] * fx / ps[2] + cx;
        const float v = ps[1] * fy / ps[2] + cy;
        const float du = xx - u, dv = yy - v;
        const float gt_hm = fmaxf(4.0f - sqrtf(du * du + dv * dv), 0.0f) / 4.0f;
        // 3D offsets (:338-346)
        const float ox = (ps[0] - cx0) / 100.0f - X;
        const float oy = (ps[1] - cy0) / 100.0f - Y;
        const float oz = (ps[2] - cz0) / 100.0f - Z;
        const float dist = sqrtf((ox * ox + oy * oy) + oz * oz);
        const float gt_hm3 = fmaxf((0.8f - dist) / 0.8f, 0.0f);
        const float d3 = 0.8f - gt_hm3 * 0.8f;
        const bool near = d3 < (0.8f - 1e-2f);
        const float ux = near ? ox / d3 : 0.0f, uy = near ? oy / d3 : 0.0f, uz = near ? oz / d3 : 0.0f;
        for (int s = 0; s < p.S; ++s) {
            const long m = i;
            const float e1 = p.hm[s].p[m * p.hm[s].cs + p.hm[s].coff + j] - gt_hm;
            const float e2 = p.hm3[s].p[m * p.hm3[s].cs + p.hm3[s].coff + j] - gt_hm3;
            const float* um = p.um[s].p + m * p.um[s].cs + p.um[s].coff + 3 * j;
            const float e3 = um[0] - ux, e4 = um[1] - uy, e5 = um[2] - uz;
            p.dhm[s].p[m * p.dhm[s].cs + p.dhm[s].coff + j] = e1;
            p.dhm3[s].p[m * p.dhm3[s].cs + p.dhm3[s].coff + j] = e2;
            float* dum = p.dum[s].p + m * p.dum[s].cs + p.dum[s].coff + 3 * j;
            dum[0] = e3; dum[1] = e4; dum[2] = e5;
            l_hm += 0.5 * (double)e1 * e1;
            l_hm3 += 0.5 * (double)e2 * e2;
            l_um += 0.5 * ((double)e3 * e3 + (double)e4 * e4 + (double)e5 * e5);
        }
    }
    // block reduction: wave shuffle, then 4 waves through LDS
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        l_hm += __shfl_xor(l_hm, m);
        l_hm3 += __shfl_xor(l_hm3, m);
        l_um += __shfl_xor(l_um, m);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = l_hm; red[1][wave] = l_hm3; red[2][wave] = l_um; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const double s = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        // [3][gridDim.x] partial rows, summed in index order by losses_out_kernel
        p.acc[(long)threadIdx.x * gridDim.x + blockIdx.x] = s;
    }
}

// L2 regulariser (losses.py:56-72): one block per weight segment; value -> acc[3], gradient wd*w -> grad
struct RegSeg { long off; long n; float wd; };
__global__ __launch_bounds__(256) void reg_loss_kernel(const float* param, const RegSeg* segs, double* acc) {
    __shared__ double red[4];
    const RegSeg sg = segs[blockIdx.x];
    double a = 0.0;
    for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < sg.n; i += (long)gridDim.y * 256) {
        const double w = param[sg.off + i];
        a += w * w;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) acc[(long)blockIdx.y * gridDim.x + blockIdx.x] = (double)sg.wd * 0.5 * (red[0] + red[1] + red[2] + red[3]);
}
// times: the micro-steps this sweep stands for (micro-batch groups: each of them carries the regulariser once)
__global__ __launch_bounds__(256) void reg_grad_kernel(const float* param, float* grad, const RegSeg* segs, int nseg, float times) {
    for (int s = blockIdx.y; s < nseg; s += gridDim.y) {
        const RegSeg sg = segs[s];
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < sg.n; i += (long)gridDim.x * 256)
            grad[sg.off + i] += times * (sg.wd * param[sg.off + i]);
    }
}
// Sums the partial rows of the loss kernels in a fixed order (1168 same-address fp64 atomics cost 100 us here and made
// the reported loss depend on their order).  loss_part = [3][n_loss] rows of loss_kernel, reg_part = n_reg partials.
// Micro-batch groups: workgroup g writes out[4 g ..] from its group's n_loss / gridDim.x consecutive rows (the loss kernel's
// threads are ordered by pixel, a group's pixels fill whole blocks); the regulariser is the same for every group.
__global__ __launch_bounds__(256) void losses_out_kernel(const double* loss_part, int n_loss, const double* reg_part, int n_reg,
                                                         float* out) {
    // one wave per loss term (the regulariser's ~9000 partials were a 146-deep chain of dependent loads on one wave: 38 us);
    // per wave: lane-strided loads, four independent accumulators, fixed shuffle tree -- the order never depends on timing
    const int lane = threadIdx.x & 63, t = threadIdx.x >> 6;
    const int bpg = n_loss / (int)gridDim.x;
    const double* src = t < 3 ? loss_part + (long)t * n_loss + (long)blockIdx.x * bpg : reg_part;
    const int n = t < 3 ? bpg : n_reg;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int i = lane;
    for (; i + 3 * 64 < n; i += 4 * 64) { a0 += src[i]; a1 += src[i + 64]; a2 += src[i + 128]; a3 += src[i + 192]; }
    for (; i < n; i += 64) a0 += src[i];
    double a = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m);
    if (lane == 0) out[4 * blockIdx.x + t] = (float)a;
}

// ------------------------------------------------------------------------------------------------
// a[i] += b[i]; b[i] = 0 -- the second micro-step slot's accumulated gradient folded into the flat accumulator (pipeline.inc)
__global__ __launch_bounds__(256) void grad_merge_kernel(float* a, float* b, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        a[i] += b[i];
        b[i] = 0.f;
    }
}

// Fused accumulate-average / clip / Adam over the flat buffers (train_single_gpu.py:86-89).
//   g = clip(acc/div, +-clip) ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; theta -= lr_t m/(sqrt(v)+eps)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* param, const float* grad, float* m, float* v, long n, float div,
                                                   float clip, float lr_t, float b1, float b2, float eps) {
#pragma clang fp contract(off)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float g = grad[i] / div;
        g = fminf(fmaxf(g, -clip), clip);
        const float mi = b1 * m[i] + (1.0f - b1) * g;
        const float vi = b2 * v[i] + (1.0f - b2) * g * g;
        m[i] = mi;
        v[i] = vi;
        param[i] = param[i] - lr_t * mi / (sqrtf(vi) + eps);
    }
}

}  // namespace dr
