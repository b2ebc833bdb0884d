/* densereg.h -- C ABI of libdensereg_hip.so: the MI355X (gfx950) dense-regression hand-pose engine.
 *
 * The reference (melonwan/denseReg) has no native interface: its hot path is a Python callable
 * contract inside a TF-1.3 graph.  Each entry point below names the reference interface it
 * replaces (paths relative to the reference tree):
 *
 *   network plug-in    network/um_v1.py:71        detect_net(dm, cfgs, coms, num_jnt, is_training)
 *                      loaded through importlib at model/hourglass_um_crop_tiny.py:863-867
 *   vote               model/hourglass_um_crop_tiny.py:743   _xyz_estimation(...) preceded by
 *                      _resume_om (:457) and followed by unnorm_xyz_pose (:462)
 *   loss               model/hourglass_um_crop_tiny.py:323   JointDetectionModel.loss
 *   optimizer          model/train_single_gpu.py:71-89,144-150  accumulate / clip / Adam apply
 *   variables          network/slim/variables.py:247  (names: <scope>/Conv_k/weights, .../BatchReNorm/...)
 *
 * Conventions
 *   - every function returns 0 on success, a negative DR_E_* code otherwise; dr_last_error() gives
 *     the message (per handle; pass NULL for errors of dr_create).
 *   - one handle per (process, device); calls on one handle are NOT re-entrant.
 *   - all *_dev pointers are device (HBM) pointers owned by the caller; tensors are dense NHWC
 *     float32 exactly as the reference feeds/fetches them.  All work is enqueued on `stream`
 *     (a hipStream_t; NULL = the null stream); nothing synchronises unless stated.
 *   - the library owns weights, workspaces, saved activations, gradient and Adam buffers.
 */
#ifndef DENSEREG_H_
#define DENSEREG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DR_ABI_VERSION 1

enum {
    DR_OK = 0,
    DR_E_INVALID = -1,      /* bad argument (null pointer, batch > max_batch, unknown name ...) */
    DR_E_UNSUPPORTED = -2,  /* in_hw not in {128,256,512}: um_v1.py:106-107 raises ValueError */
    DR_E_STATE = -3,        /* call order (params not finalised, handle not created for training) */
    DR_E_DEVICE = -4,       /* HIP runtime error */
    DR_E_NOMEM = -5
};

typedef struct dr_handle dr_handle;
typedef void* dr_stream;    /* hipStream_t */

/* Flags the reference reads as process globals (hourglass_um_crop_tiny.py:29-60) + sizing. */
typedef struct dr_config {
    int32_t num_stack;      /* --num_stack   (default 2)   */
    int32_t num_fea;        /* --num_fea / --fea_num (128) */
    int32_t num_jnt;        /* dataset.jnt_num: icvl 16, nyu 14, msra 21 */
    int32_t in_hw;          /* input crop side, 128 (256/512 accepted by um_v1.py:99-104) */
    int32_t kernel_size;    /* --kernel_size (3): residual kxk conv and hourglass pool */
    int32_t max_batch;      /* largest B any call will pass */
    int32_t device;         /* HIP device ordinal */
    int32_t training;       /* 0: inference only; 1: also allocate saved activations/grads/Adam */
} dr_config;

int dr_abi_version(void);
const char* dr_backend(void);                     /* "hip-gfx950" for the product library */
int dr_create(const dr_config* cfg, dr_handle** out);
void dr_destroy(dr_handle* h);
const char* dr_last_error(const dr_handle* h);

/* ---- variables (TF names, HWIO weights; host float32 buffers) ------------------------------- */
int dr_param_count(const dr_handle* h);
int dr_param_info(const dr_handle* h, int index, const char** name, int32_t dims[4], int32_t* ndim,
                  int32_t* trainable);
int dr_load_param(dr_handle* h, const char* name, const float* host, size_t count);
int dr_read_param(dr_handle* h, const char* name, float* host, size_t count);
/* Both also accept, on a training handle, the zero-debias slot variables of the moving statistics that
 * assign_moving_average keeps ([TF1.3-semantics]): "<scope>/BatchReNorm/moving_{mean,variance}/biased" (Cout
 * floats) and ".../local_step" (1 float).  Load them AFTER the moving statistic itself (which resets them). */
/* Pack weights for the MFMA kernels, fold eval-mode BatchReNorm; call after loading, before any
 * forward.  Synchronises. */
int dr_finalize_params(dr_handle* h, dr_stream stream);

/* CRC-32C (Castagnoli) continuation: dr_crc32c(0, data, n) for a fresh checksum.  Host code; the checksum of
 * TensorFlow tensor-bundle checkpoints (densereg_amd/checkpoint.py). */
uint32_t dr_crc32c(uint32_t crc, const void* data, size_t n);

/* ---- input front-end: raw frame -> crop + centre of mass, augmentation ------------- */
/* All pointers are device pointers; these three entry points need no handle (no weights, no workspace) and run
 * on the current device.  One launch per call, nothing synchronises.
 *
 * data/preprocess.py:10-79 crop_from_xyz_pose + :131-142 center_of_mass.  frames (B,H,W) depth in mm, pose (B,3J)
 * xyz in mm (ground truth at training time, the previous estimate when tracking), cfg (B,6) = fx,fy,cx,cy,w,h of
 * the frame.  Box = projected joints +- pad pixels, cropped, zero padded to a square, bilinear-resized
 * (tf.image.resize_images defaults) to out_hw x out_hw, background removed (icvl != 0: depth >= 500 -> 0, else
 * depth >= min joint-pixel depth + 250 -> 0).  Outputs: crops (B,out_hw,out_hw), new_cfg (B,6) the camera of the
 * crop, com (B,3) = mean positive depth at the centre pixel of the crop, back-projected. */
int dr_crop_from_pose(int B, const float* frames_dev, int H, int W, const float* pose_mm_dev, int J, const float* cfg_dev,
                      int icvl, float pad, int out_hw, float* crops_dev, float* new_cfg_dev, float* com_dev,
                      dr_stream stream);
/* data/preprocess.py:81-129 crop_from_bbx + center_of_mass: bbx (B,5) = top,left,bottom,right,depth threshold. */
int dr_crop_from_bbx(int B, const float* frames_dev, int H, int W, const float* bbx_dev, const float* cfg_dev, int out_hw,
                     float* crops_dev, float* new_cfg_dev, float* com_dev, dr_stream stream);
/* data/preprocess.py:234-268 data_aug: rotate about the image centre (nearest), rescale rows/columns by
 * ratio_h/ratio_w (nearest), centre crop-or-pad back to (H,W); the pose follows in uvd space around the centre of
 * mass.  draws (B,3) = angle [rad], ratio_h, ratio_w -- the reference draws U(-pi,pi) and clip(N(1,.2),.9,1.1);
 * the caller supplies them (the Python host draws them, parity tests inject them). */
int dr_data_aug(int B, const float* dms_dev, int H, int W, const float* pose_mm_dev, int J, const float* cfg_dev,
                const float* com_dev, const float* draws_dev, float* out_dms_dev, float* out_pose_dev, dr_stream stream);

/* ---- pre-processing on the path --------------------------------------------------------------- */
/* data/preprocess.py:176-187 norm_dm: dm_mm (B,hw,hw,1), com (B,3) -> dm_norm (B,hw,hw,1) */
int dr_norm_dm(dr_handle* h, int B, const float* dm_mm_dev, const float* com_dev, float* dm_norm_dev,
               dr_stream stream);

/* ---- inference ---------------------------------------------------------------------------------- */
/* detect_net(..., is_training=False); fills the LAST stack's maps (hourglass_um_crop_tiny.py:451-455)
 * when the pointers are non-NULL: hm (B,h,w,J), hm3 (B,h,w,J), um (B,h,w,3J), h = w = in_hw/4. */
int dr_forward_eval(dr_handle* h, int B, const float* dm_norm_dev, float* hm_dev, float* hm3_dev,
                    float* um_dev, dr_stream stream);
/* Maps of any stack of the most recent forward (eval or train), same layouts. */
int dr_read_maps(dr_handle* h, int B, int stack, float* hm_dev, float* hm3_dev, float* um_dev,
                 dr_stream stream);
/* _resume_om + _xyz_estimation + unnorm_xyz_pose on caller-provided maps.
 * dm_norm (B,hw,hw,1), cfg (B,6)=fx,fy,cx,cy,w,h, com (B,3) -> xyz_mm (B,3J). */
int dr_vote(dr_handle* h, int B, const float* hm_dev, const float* hm3_dev, const float* um_dev,
            const float* dm_norm_dev, const float* cfg_dev, const float* com_dev, float* xyz_mm_dev,
            dr_stream stream);
/* JointDetectionModel.test (:442-462): forward(eval) + vote without materialising dense maps. */
int dr_infer(dr_handle* h, int B, const float* dm_norm_dev, const float* cfg_dev, const float* com_dev,
             float* xyz_mm_dev, dr_stream stream);

/* ---- training (handle created with training=1) -------------------------------------------------- */
enum { DR_DROPOUT_OFF = 0, DR_DROPOUT_MASK = 1, DR_DROPOUT_RNG = 2 };
/* detect_net(..., is_training=True): batch-statistics BatchReNorm (updates moving stats, r_max,
 * d_max, curr_t: slim/ops.py:130-171), dropout 0.5 after the two 512-wide head convs.
 * dropout_mode MASK: keep_mask_dev = uint8 [num_stack][2][B*h*w*512] (1 = keep, scaled x2);
 * RNG: counter-based generator keyed by `seed`. */
int dr_forward_train(dr_handle* h, int B, const float* dm_norm_dev, int dropout_mode,
                     const uint8_t* keep_mask_dev, uint64_t seed, dr_stream stream);
/* JointDetectionModel.loss (:323-371) on the maps of the last dr_forward_train: synthesises the
 * targets (_hm_2d/_hm_3d/_um), writes losses_dev[4] = {hm, hm3, um, reg} (sum over stacks,
 * l2_loss = sum(x^2)/2) and seeds the gradients of all stacks' maps. */
int dr_loss(dr_handle* h, int B, const float* dm_norm_dev, const float* pose_mm_dev, const float* cfg_dev,
            const float* com_dev, float* losses_dev, dr_stream stream);
/* Back-propagation through the whole graph; ADDS d(total loss)/d(theta), including the L2
 * regulariser term, into the flat gradient accumulator (train_single_gpu.py:84 accum_op). */
int dr_backward(dr_handle* h, int B, dr_stream stream);
int dr_zero_grad(dr_handle* h, dr_stream stream);                       /* reset_op (:83) */
/* Micro-steps in flight (training handles; default 1).  The reference accumulates `sub_batch` micro-steps on the same weights
 * between two optimizer steps (train_single_gpu.py:138-150); consecutive micro-steps depend on each other only through the
 * BatchReNorm moving statistics (forward k+1 reads what forward k wrote, slim/ops.py:134-162) and the gradient sum.  With
 * depth 2 the handle owns two sets of per-micro-step buffers and two library streams: dr_forward_train / dr_loss / dr_backward
 * of micro-step k are enqueued on set k % 2's stream (after whatever `stream` holds at the time of the call), forward k+1
 * starts when forward k has finished, and the two kernel streams overlap.  The crops of dr_forward_train and the poses / camera
 * parameters / centres of mass of dr_loss are copied in `stream`'s order at the time of the call (the caller may recycle those
 * buffers right behind the call, on `stream`); a DR_DROPOUT_MASK keep mask is NOT copied (it is a test hook of 2 KB per map pixel):
 * it must stay untouched until `stream` has been ordered behind the micro-step's dr_backward (dr_sync_grads, dr_zero_grad,
 * dr_apply_adam).  dr_loss orders `stream` behind the loss kernels when losses_dev is given (it is valid in stream order);
 * dr_zero_grad, dr_sync_grads and dr_apply_adam order `stream` behind every micro-step in flight.  A rejected call (DR_E_INVALID /
 * DR_E_UNSUPPORTED from dr_forward_train) leaves the handle where it was: the previous micro-step can still be continued.  Each set accumulates its own gradient, summed in a fixed order by
 * dr_sync_grads / dr_apply_adam: results are deterministic, and equal depth 1's up to the rounding of that one addition.
 * Entry points that read the handle's buffers from the host (dr_read_param, dr_read_activation, ...) drain the pipeline first. */
int dr_set_pipeline(dr_handle* h, int depth);
/* Micro-batch groups (training handles; default 1, at most 8).  With groups = G the next dr_forward_train / dr_loss /
 * dr_backward calls take the G micro-batches of one accumulation window at once: their B crops (and the rows of pose / cfg /
 * com / a keep mask) are G consecutive micro-batches of B/G crops, and one pass of launches does what G micro-steps of the
 * reference do one after the other (train_single_gpu.py:138-150) -- BatchReNorm statistics, r / d and the clip schedule per
 * micro-batch; the moving statistics and the zero-debias accumulators chained through the micro-batches in order (micro-batch
 * g normalises with what g-1 left, slim/ops.py:134-162); the gradient sum and G times the regulariser added to the accumulator.
 * losses_dev then receives G rows of four, one per micro-batch.  Same numbers as G separate micro-steps up to the rounding
 * of sums taken in another order (the kernels see 4x-8x the rows per launch: that is the point).  DR_DROPOUT_RNG draws one
 * mask over all B crops from `seed`.  Requires B % G == 0 and B/G a multiple of 8 crops (whole 128-row tiles per micro-batch
 * in the smallest, 4x4, layers); DR_E_UNSUPPORTED otherwise. */
int dr_set_groups(dr_handle* h, int groups);
/* Orders `stream` behind every micro-step in flight and folds all sets' accumulated gradients into the buffer dr_flat_grad
 * names (call it before an all-reduce of that buffer: train_multi_gpu.py:16-39).  A no-op at depth 1. */
int dr_sync_grads(dr_handle* h, dr_stream stream);
/* Flat fp32 views in TF trainable-variable creation order (for RCCL all-reduce / checkpoints). */
int dr_flat_grad(dr_handle* h, float** dev_ptr, size_t* count);
int dr_flat_param(dr_handle* h, float** dev_ptr, size_t* count);
/* Adam's first / second moment estimates, same layout as the flat parameter buffer (TF slot variables
 * "<var>/Adam" and "<var>/Adam_1"; checkpoint import/export, resume). */
int dr_flat_adam(dr_handle* h, float** m_dev_ptr, float** v_dev_ptr, size_t* count);
/* g = clip(acc / div, -clip, clip); Adam(beta1=0.5, beta2=0.999, eps=1e-8), TF update rule;
 * step is 1-based (train_single_gpu.py:86-89; hourglass_um_crop_tiny.py:436-439).  Re-packs the
 * weights for the next forward. */
int dr_apply_adam(dr_handle* h, float lr, float div, float clip, int64_t step, dr_stream stream);

/* ---- dataset formats: PNG depth frames out of the TFRecord files --------------------------- */
/* Host: undo the PNG row filters (PNG spec 9.2) of an inflated IDAT stream.  `filtered` = height rows of
 * (1 filter-type byte + row_bytes), bpp = bytes per pixel (3 for NYU's RGB8, 2 for 16-bit grey); out = height*row_bytes
 * bytes.  Replaces the filter stage of tf.image.decode_png (data/nyu.py:148-149, data/icvl.py, data/msra.py:190).
 * DR_E_INVALID on an unknown filter type. */
int dr_png_unfilter(const uint8_t* filtered, int height, int row_bytes, int bpp, uint8_t* out);
/* Device: PNG samples -> fp32 depth frame in mm.  DR_SAMPLES_RGB8_GB: 3 bytes per pixel, depth = (G << 8) | B
 * (data/nyu.py:151-156); DR_SAMPLES_GREY16_BE: 2 bytes per pixel, big-endian (decode_png(dtype=uint16) + to_float).
 * samples_dev 4-byte aligned, depth_dev 16-byte aligned, npix pixels (any number of frames back to back). */
#define DR_SAMPLES_RGB8_GB 0
#define DR_SAMPLES_GREY16_BE 1
int dr_depth_from_samples(const uint8_t* samples_dev, long npix, int mode, float* depth_dev, dr_stream stream);

/* ---- precision ------------------------------------------------------------------------------------ */
/* Matrix-core arithmetic of the convolutions of dr_forward_eval / dr_infer (BASELINE config 5 asks for a bf16 MFMA
 * conv path): DR_PREC_F32 (default) = v_mfma_f32_32x32x2_f32; DR_PREC_BF16 = activations and weights rounded to bf16
 * (nearest even) as they enter the matrix cores, fp32 accumulation, fp32 tensors, epilogues, heads' outputs and vote.
 * On a training handle the train-mode forward, the input-gradient and the weight-gradient convolutions follow
 * (v_mfma_f32_32x32x16_bf16 in all three); BatchReNorm arithmetic, loss, gradients, Adam and the master weights stay fp32.
 * Four kinds of INTERNAL tensors of the training step are stored as bf16 there, each read back only by kernels that would round
 * it to bf16 anyway or normalise exactly the stored values: an activation whose only reader is a convolution and its gradient,
 * the gradient with respect to a BatchReNorm layer's raw output, and that raw output itself (the layer's statistics are sums
 * over the stored values).  Inputs, outputs, parameters and parameter gradients of this ABI are fp32 in both precisions.
 * Call before dr_finalize_params: changing the precision un-finalizes the handle because the packed weights change type. */
#define DR_PREC_F32 0
#define DR_PREC_BF16 1
int dr_set_precision(dr_handle* h, int precision);

/* ---- eval-mode fusion ---------------------------------------------------------------------------- */
/* In eval mode (dr_forward_eval / dr_infer, fp32 matrix cores, num_fea a multiple of 32 up to 128) the part of every hourglass
 * below 16x16 pixels -- 24 convolutions, 3 pools, 2 upsample-adds of network/um_v1.py:51-69 -- runs as ONE launch with its
 * intermediate tensors in LDS (on by default; same results up to fp32 summation order).  on = 0 launches every op on its own, which
 * also keeps every layer's output in HBM for dr_read_activation. */
int dr_set_fusion(dr_handle* h, int on);

/* ---- introspection (tests / profiling) ----------------------------------------------------------- */
/* Post-activation output of conv `scope` (e.g. "Conv_12") of the last forward as dense NHWC.  After an fp32 dr_forward_train, for a
 * BatchReNorm layer: "<scope>#raw" = its output before BatchReNorm (count = B*H*W*cout), "<scope>#fold" = [scale | shift] (count =
 * 2*cout), the multiply-add that forward applied to it (micro-batch group 0) -- together what every ReLU of that forward decided. */
int dr_read_activation(dr_handle* h, const char* scope, int B, float* host, size_t count);
/* Algorithmic conv FLOPs per crop (forward), SURVEY section 8(d). */
double dr_conv_flops_per_crop(const dr_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* DENSEREG_H_ */
