// frontend.h -- raw depth frame -> network input and the training-time augmentation (row 3).
//
// Replaces, per frame, the TF op chains of the reference's input pipeline:
//   crop_from_xyz_pose / crop_from_bbx   data/preprocess.py:10-129  (box, crop, zero pad to a square, bilinear
//                                        resize, background threshold, camera of the crop)
//   center_of_mass                       data/preprocess.py:131-142
//   data_aug                             data/preprocess.py:234-268  (rotate, anisotropic nearest rescale, centre
//                                        crop-or-pad; the same transform on the pose)
// One workgroup per frame.  Nothing is materialised between the steps: an output pixel walks the chain of
// coordinate maps backwards and reads the source frame directly.  Arithmetic is fp32 in the reference's op order
// with contraction off (oracle/frontend.py states the TensorFlow kernel semantics this follows).
// HBM-bound: a crop reads at most the box (<= H*W*4 B) and writes out*out*4 B.
#pragma once
#include "dr_platform.h"

namespace dr {

struct CropParams {
    const float* frames; int H, W;             // [B][H][W] depth in mm
    const float* pose; int J;                  // [B][3J] xyz in mm (box from pose) or nullptr
    const float* bbx;                          // [B][5] top,left,bottom,right,d_th (box given) or nullptr
    const float* cfg;                          // [B][6] fx,fy,cx,cy,w,h
    int icvl; float pad; int out_hw;
    float* crops; float* new_cfg; float* com;  // [B][out][out], [B][6], [B][3]
};

__global__ __launch_bounds__(256) void crop_com_kernel(const CropParams p) {
#pragma clang fp contract(off)
    __shared__ int s_box[7];                   // top, left, h, w, longer, off_h, off_w
    __shared__ float s_dth;
    __shared__ double s_sum[256];
    __shared__ int s_cnt[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* dm = p.frames + (long)b * p.H * p.W;
    const float* cfg = p.cfg + b * 6;
    if (tid == 0) {
        int top, left, bottom, right;
        float d_th;
        if (p.bbx) {
            const float* bb = p.bbx + b * 5;
            top = (int)bb[0]; left = (int)bb[1]; bottom = (int)bb[2]; right = (int)bb[3];
            d_th = bb[4];
        } else {
            float mnu = 3.4e38f, mnv = 3.4e38f, mxu = -3.4e38f, mxv = -3.4e38f, dmin = 3.402823466e38f;
            for (int j = 0; j < p.J; ++j) {
                const float* q = p.pose + ((long)b * p.J + j) * 3;
                const float u = q[0] * cfg[0] / q[2] + cfg[2];
                const float v = q[1] * cfg[1] / q[2] + cfg[3];
                mnu = fminf(mnu, u); mxu = fmaxf(mxu, u); mnv = fminf(mnv, v); mxv = fmaxf(mxv, v);
                int uu = (int)u, vv = (int)v;
                uu = uu < 0 ? 0 : (uu > p.W - 1 ? p.W - 1 : uu);
                vv = vv < 0 ? 0 : (vv > p.H - 1 ? p.H - 1 : vv);
                const float d = dm[vv * p.W + uu];
                if (d > 100.f) dmin = fminf(dmin, d);
            }
            const float pad = p.pad, fh = cfg[5], fw = cfg[4];
            const float t = fminf(fmaxf(mnv - pad, 0.f), fh - 2 * pad);
            const float l = fminf(fmaxf(mnu - pad, 0.f), fw - 2 * pad);
            const float bt = fmaxf(fminf(mxv + pad, fh), t + 2 * pad - 1);
            const float r = fmaxf(fminf(mxu + pad, fw), l + 2 * pad - 1);
            top = (int)t; left = (int)l; bottom = (int)bt; right = (int)r;
            d_th = p.icvl ? 500.f : (dmin == 3.402823466e38f ? dmin : dmin + 250.f);
        }
        // A box is data (dataset records, nyu_bbx.pkl): the reference's crop_to_bounding_box raises on a box that leaves
        // the frame or has no area.  Here pixels outside the frame read as background (0, like the square's zero padding,
        // see sq below) and a degenerate box is treated as one pixel wide, so the crop camera stays finite.
        int h = bottom - top, w = right - left;
        h = h < 0 ? 0 : h; w = w < 0 ? 0 : w;
        int longer = h > w ? h : w;
        longer = longer < 1 ? 1 : longer;
        const int off_h = (int)((double)(longer - h) / 2.0), off_w = (int)((double)(longer - w) / 2.0);
        s_box[0] = top; s_box[1] = left; s_box[2] = h; s_box[3] = w; s_box[4] = longer; s_box[5] = off_h; s_box[6] = off_w;
        s_dth = d_th;
        const float rx = (float)((double)longer / (double)p.out_hw), ry = rx;
        float* nc = p.new_cfg + b * 6;
        nc[0] = cfg[0] / rx; nc[1] = cfg[1] / ry;
        nc[2] = (cfg[2] - (float)left + (float)off_w) / rx;
        nc[3] = (cfg[3] - (float)top + (float)off_h) / ry;
        nc[4] = (float)p.out_hw; nc[5] = (float)p.out_hw;
    }
    __syncthreads();
    const int top = s_box[0], left = s_box[1], bh = s_box[2], bw = s_box[3], longer = s_box[4], off_h = s_box[5], off_w = s_box[6];
    const float d_th = s_dth;
    const int out = p.out_hw;
    const float scale = (float)longer / (float)out;
    // sample of the virtual zero-padded square at integer (y, x)
    auto sq = [&](int y, int x) -> float {
        const int cy = y - off_h, cx = x - off_w;
        if (cy < 0 || cy >= bh || cx < 0 || cx >= bw) return 0.f;
        const int fy = top + cy, fx = left + cx;
        if (fy < 0 || fy >= p.H || fx < 0 || fx >= p.W) return 0.f;          // box beyond the frame: background
        return dm[(long)fy * p.W + fx];
    };
    double sum = 0.0;
    int cnt = 0;
    float* dst = p.crops + (long)b * out * out;
    for (int i = tid; i < out * out; i += 256) {
        const int oy = i / out, ox = i % out;
        const float ys = (float)oy * scale, xs = (float)ox * scale;
        const int y0 = (int)ys, x0 = (int)xs;
        int y1 = (int)ceilf(ys), x1 = (int)ceilf(xs);
        y1 = y1 < longer - 1 ? y1 : longer - 1;
        x1 = x1 < longer - 1 ? x1 : longer - 1;
        const float yl = ys - (float)y0, xl = xs - (float)x0;
        const float tl = sq(y0, x0), tr = sq(y0, x1), bl = sq(y1, x0), br = sq(y1, x1);
        const float tp = tl + (tr - tl) * xl;
        const float bt = bl + (br - bl) * xl;
        float v = tp + (bt - tp) * yl;
        v = v < d_th ? v : 0.f;
        dst[i] = v;
        if (v > 0.f) { sum += (double)v; ++cnt; }
    }
    s_sum[tid] = sum; s_cnt[tid] = cnt;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { s_sum[tid] += s_sum[tid + o]; s_cnt[tid] += s_cnt[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        float ave_d = s_cnt[0] > 0 ? (float)(s_sum[0] / (double)s_cnt[0]) : 200.f;
        ave_d = fmaxf(ave_d, 200.f);
        const float* nc = p.new_cfg + b * 6;
        const float au = (float)((double)out / 2.0), av = au;
        float* c = p.com + b * 3;
        c[0] = (au - nc[2]) * ave_d / nc[0];
        c[1] = (av - nc[3]) * ave_d / nc[1];
        c[2] = ave_d;
    }
}

struct AugParams {
    const float* dms; int H, W;                // [B][H][W] crops
    const float* pose; int J;                  // [B][3J]
    const float* cfg; const float* com;        // [B][6], [B][3]
    const float* draws;                        // [B][3]: angle, ratio_h, ratio_w
    float* out_dms; float* out_pose;
};

__device__ __forceinline__ int round_half_away(float x) { return (int)(x >= 0.f ? floorf(x + 0.5f) : ceilf(x - 0.5f)); }

__global__ __launch_bounds__(256) void data_aug_kernel(const AugParams p) {
#pragma clang fp contract(off)
    const int b = blockIdx.x, tid = threadIdx.x;
    const int H = p.H, W = p.W;
    const float* dm = p.dms + (long)b * H * W;
    const float* cfg = p.cfg + b * 6;
    const float angle = p.draws[b * 3 + 0], ratio_h = p.draws[b * 3 + 1], ratio_w = p.draws[b * 3 + 2];
    const float c = cosf(angle), s = sinf(angle);
    const float ox = ((float)(W - 1) - (c * (float)(W - 1) - s * (float)(H - 1))) / 2.0f;
    const float oy = ((float)(H - 1) - (s * (float)(W - 1) + c * (float)(H - 1))) / 2.0f;
    const int th = (int)((float)H * ratio_h), tw = (int)((float)W * ratio_w);
    // resize_image_with_crop_or_pad offsets (floor division)
    auto fdiv2 = [](int a) { return a >= 0 ? a / 2 : -((-a + 1) / 2); };
    const int dh = H - th, dw = W - tw;                      // target - size
    const int ch = fdiv2(-dh) > 0 ? fdiv2(-dh) : 0, cw = fdiv2(-dw) > 0 ? fdiv2(-dw) : 0;
    const int ph = fdiv2(dh) > 0 ? fdiv2(dh) : 0, pw = fdiv2(dw) > 0 ? fdiv2(dw) : 0;
    const int vis_h = th < H ? th : H, vis_w = tw < W ? tw : W;   // rows/cols of the resized image that survive
    const float sy = (float)H / (float)th, sx = (float)W / (float)tw;
    float* dst = p.out_dms + (long)b * H * W;
    for (int i = tid; i < H * W; i += 256) {
        const int y = i / W, x = i % W;
        float v = 0.f;
        const int ry = y - ph, rx = x - pw;                   // position inside the cropped resized image
        if (ry >= 0 && ry < vis_h && rx >= 0 && rx < vis_w) {
            int yy = (int)floorf((float)(ry + ch) * sy), xx = (int)floorf((float)(rx + cw) * sx);   // nearest resize
            yy = yy < H - 1 ? yy : H - 1;
            xx = xx < W - 1 ? xx : W - 1;
            const int srcx = round_half_away(c * (float)xx - s * (float)yy + ox);                      // rotation
            const int srcy = round_half_away(s * (float)xx + c * (float)yy + oy);
            if (srcx >= 0 && srcx < W && srcy >= 0 && srcy < H) v = dm[srcy * W + srcx];
        }
        dst[i] = v;
    }
    // pose: rotate the uvd offsets from the centre of mass, rescale, back-project (preprocess.py:241-261)
    if (tid < p.J) {
        const float* cm = p.com + b * 3;
        const float ucom = cm[0] * cfg[0] / cm[2] + cfg[2], vcom = cm[1] * cfg[1] / cm[2] + cfg[3], dcom = cm[2];
        const float* q = p.pose + ((long)b * p.J + tid) * 3;
        const float u = q[0] * cfg[0] / q[2] + cfg[2] - ucom;
        const float v = q[1] * cfg[1] / q[2] + cfg[3] - vcom;
        const float d = q[2] - dcom;
        // row vector times [[c,-s,0],[s,c,0],[0,0,1]]
        float ru = u * c + v * s + d * 0.f;
        float rv = u * (-s) + v * c + d * 0.f;
        float rd = u * 0.f + v * 0.f + d * 1.f;
        ru = ru * ratio_w + ucom;
        rv = rv * ratio_h + vcom;
        rd = rd * 1.0f + dcom;
        float* o = p.out_pose + ((long)b * p.J + tid) * 3;
        o[0] = (ru - cfg[2]) * rd / cfg[0];
        o[1] = (rv - cfg[3]) * rd / cfg[1];
        o[2] = rd;
    }
}

}  // namespace dr
