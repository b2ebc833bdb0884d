// kernels_misc.h -- the HBM-bound kernels around the convolutions (forward side).
//
//   stem_conv_kernel     conv 7x7/s2 1->32 of um_v1.py:86 (Cin=1: direct, not a GEMM)
//   maxpool_kernel       tf.nn.max_pool 'SAME' k in {2,3}, s=2 (ops.py:640-669; um_v1.py:57,91)
//   upsample_add_kernel  upper1 + nearest_x2(lower3) (um_v1.py:66-69; ops.py:671-677)
//   uvd_kernel           coordinate / depth channels (um_v1.py:109-121) + dense tiny_dm
//   copy_channels_kernel channel-slice copy between NHWC views (tf.concat, um_v1.py:137-175)
//   norm_dm_kernel       data/preprocess.py:176-187
//   moments_kernel       per-channel sum / sum of squares (tf.nn.moments, ops.py:132)
#pragma once
#include "dr_platform.h"

namespace dr {

struct View {            // NHWC tensor view: element (m, c) at p[m*cs + coff + c]
    float* p; int cs; int coff; int C;
};

// ------------------------------------------------------------------------------------------
// Stem: y[b,oy,ox,n] = sum_{ky,kx} x[b, oy*s+ky-pt, ox*s+kx-pl] * w[ky,kx,0,n]
// block = 256 threads = 64 output pixels x 4 groups of 8 channels (Cout = 32)
// ------------------------------------------------------------------------------------------
struct StemParams {
    const float* x; int B, H, W;         // input (B,H,W,1)
    const float* w;                      // HWIO [k][k][1][32]
    int k, stride, pad_t, pad_l, Ho, Wo;
    float* y; int y_cs;                  // (B,Ho,Wo,32)
    const float* scale; const float* shift; int relu;
};

__global__ __launch_bounds__(256) void stem_conv_kernel(const StemParams p) {
    __shared__ float ws[7 * 7 * 32];
    const int tid = threadIdx.x;
    const int nw = p.k * p.k * 32;
    for (int i = tid; i < nw; i += 256) ws[i] = p.w[i];
    __syncthreads();
    const int pix = blockIdx.x * 64 + (tid >> 2);
    const int cg = (tid & 3) * 8;
    const int M = p.B * p.Ho * p.Wo;
    if (pix >= M) return;
    const int b = pix / (p.Ho * p.Wo);
    const int rem = pix % (p.Ho * p.Wo);
    const int oy = rem / p.Wo, ox = rem % p.Wo;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    const float* xb = p.x + (long)b * p.H * p.W;
    for (int ky = 0; ky < p.k; ++ky) {
        const int iy = oy * p.stride + ky - p.pad_t;
        if (iy < 0 || iy >= p.H) continue;
        for (int kx = 0; kx < p.k; ++kx) {
            const int ix = ox * p.stride + kx - p.pad_l;
            if (ix < 0 || ix >= p.W) continue;
            const float xv = xb[iy * p.W + ix];
            const float* wr = &ws[(ky * p.k + kx) * 32 + cg];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = fmaf(xv, wr[c], acc[c]);
        }
    }
    float* yo = p.y + (long)pix * p.y_cs + cg;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float v = acc[c];
        const float sc = p.scale ? p.scale[cg + c] : 1.f;
        const float sh = p.shift ? p.shift[cg + c] : 0.f;
        v = v * sc + sh;
        if (p.relu) v = fmaxf(v, 0.f);
        yo[c] = v;
    }
}

// ------------------------------------------------------------------------------------------
// Max pool, stride 2, TF SAME (pad only bottom/right for the sizes on this path; padding never
// wins).  Thread = (output pixel, 4 channels).
// ------------------------------------------------------------------------------------------
// arg (nullable, training): [B*Ho*Wo][C] bytes, the window position ky*k+kx of the FIRST maximum in scan order -- what the
// backward pass gathers by (no floating-point atomics, no clearing of the input gradient).
__global__ __launch_bounds__(256) void maxpool_kernel(const float* x, int x_cs, int x_coff, int B, int H, int W,
                                                      int C, int k, int pad_t, int pad_l, float* y, int y_cs,
                                                      int y_coff, int Ho, int Wo, unsigned char* arg) {
    const int c4n = C / 4;
    const long total = (long)B * Ho * Wo * c4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = int(i % c4n);
        const long pix = i / c4n;
        const int ox = int(pix % Wo);
        const int oy = int((pix / Wo) % Ho);
        const int b = int(pix / ((long)Wo * Ho));
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        int ax = -1, ay = -1, az = -1, aw = -1;                      // first maximum per channel (strict >; the first valid tap seeds it)
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * 2 + ky - pad_t;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * 2 + kx - pad_l;
                if (ix < 0 || ix >= W) continue;
                const float4 v = *reinterpret_cast<const float4*>(x + ((long)(b * H + iy) * W + ix) * x_cs + x_coff + c4 * 4);
                const int t = ky * k + kx;
                if (v.x > m.x || ax < 0) { m.x = v.x; ax = t; }
                if (v.y > m.y || ay < 0) { m.y = v.y; ay = t; }
                if (v.z > m.z || az < 0) { m.z = v.z; az = t; }
                if (v.w > m.w || aw < 0) { m.w = v.w; aw = t; }
            }
        }
        *reinterpret_cast<float4*>(y + pix * y_cs + y_coff + c4 * 4) = m;
        if (arg) *reinterpret_cast<unsigned*>(arg + pix * C + c4 * 4) = (unsigned)ax | ((unsigned)ay << 8) | ((unsigned)az << 16) | ((unsigned)aw << 24);
    }
}

// out[b,y,x,c] = a[b,y,x,c] + lo[b,y/2,x/2,c]        (H,W = full-res dims)
__global__ __launch_bounds__(256) void upsample_add_kernel(const float* a, int a_cs, int a_coff, const float* lo,
                                                           int lo_cs, int lo_coff, float* out, int o_cs, int o_coff,
                                                           int B, int H, int W, int C) {
    const int c4n = C / 4;
    const long total = (long)B * H * W * c4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = int(i % c4n);
        const long pix = i / c4n;
        const int xx = int(pix % W);
        const int yy = int((pix / W) % H);
        const int b = int(pix / ((long)W * H));
        const long lpix = ((long)b * (H / 2) + yy / 2) * (W / 2) + xx / 2;
        const float4 u = *reinterpret_cast<const float4*>(a + pix * a_cs + a_coff + c4 * 4);
        const float4 v = *reinterpret_cast<const float4*>(lo + lpix * lo_cs + lo_coff + c4 * 4);
        *reinterpret_cast<float4*>(out + pix * o_cs + o_coff + c4 * 4) =
            make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
}

// uvd channels at map resolution (h = w = in_hw/4) + dense tiny_dm (B*h*w)
__global__ __launch_bounds__(256) void uvd_kernel(const float* dm, int B, int in_hw, float* tiny, float* d0, int d0_cs,
                                                  int d0_coff, float* d1, int d1_cs, int d1_coff) {
    const int h = in_hw / 4;
    const int total = B * h * h;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int xx = i % h, yy = (i / h) % h, b = i / (h * h);
        const float t = dm[((long)b * in_hw + yy * 4) * in_hw + xx * 4];      // bicubic /4 == [::4, ::4]
        const float uu = (float)xx / (float)(h / 2) - 1.0f;
        const float vv = (float)yy / (float)(h / 2) - 1.0f;
        tiny[i] = t;
        if (d0) { float* q = d0 + (long)i * d0_cs + d0_coff; q[0] = uu; q[1] = vv; q[2] = t; }
        if (d1) { float* q = d1 + (long)i * d1_cs + d1_coff; q[0] = uu; q[1] = vv; q[2] = t; }
    }
}

// dst(m, c) (+)= src(m, c) for c < C
// Zero several buffers in ONE launch: segment y of the table = floats [off, off + n_per_b * B) of `base` (16-byte
// aligned, multiples of 4).  The backward sweep needs ~24 gradient buffers cleared per step (plan_backward); as
// separate memsets that was 24 x 5.7 us of serial launches in front of the loss kernel.
struct ZeroSeg { long off; long n_per_b; };
__global__ __launch_bounds__(256) void zero_segments_kernel(float* base, const ZeroSeg* segs, int B) {
    const ZeroSeg sg = segs[blockIdx.y];
    float4* p = reinterpret_cast<float4*>(base + sg.off);
    const long n4 = sg.n_per_b * B / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// bf16-stored source (element stride s_cs, as the BatchReNorm apply pass of the bf16 path leaves a single-reader activation) -> dense
// fp32: what dr_read_activation hands out for such a tensor
__global__ __launch_bounds__(256) void copy_channels_from_bf16_kernel(const __bf16* src, int s_cs, int s_coff, float* dst, int d_cs,
                                                                      int d_coff, long M, int C) {
    const long total = M * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = int(i % C);
        const long m = i / C;
        dst[m * d_cs + d_coff + c] = (float)src[m * s_cs + s_coff + c];
    }
}

__global__ __launch_bounds__(256) void copy_channels_kernel(const float* src, int s_cs, int s_coff, float* dst, int d_cs,
                                                            int d_coff, long M, int C, int accumulate) {
    if (((C | s_cs | s_coff | d_cs | d_coff) & 3) == 0) {                 // uniform: 16-byte rows on both sides
        const int c4n = C / 4;
        const long total = M * c4n;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            const int c = int(i % c4n) * 4;
            const long m = i / c4n;
            const float4 v = *reinterpret_cast<const float4*>(src + m * s_cs + s_coff + c);
            float4* q = reinterpret_cast<float4*>(dst + m * d_cs + d_coff + c);
            if (accumulate) {
                const float4 o = *q;
                *q = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
            } else {
                *q = v;
            }
        }
        return;
    }
    const long total = M * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = int(i % C);
        const long m = i / C;
        const float v = src[m * s_cs + s_coff + c];
        float* q = dst + m * d_cs + d_coff + c;
        *q = accumulate ? (*q + v) : v;
    }
}

// data/preprocess.py:176-187
__global__ __launch_bounds__(256) void norm_dm_kernel(const float* dm, const float* com, float* out, int B, int npix) {
    const long total = (long)B * npix;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = int(i / npix);
        const float cz = com[b * 3 + 2];
        const float max_depth = cz + 300.0f * 0.5f;
        const float min_depth = cz - 300.0f * 0.5f;
        const float d = dm[i];
        const bool m = (d < max_depth) && (d > (min_depth - 300.0f * 0.5f));
        out[i] = m ? (d - min_depth) / 300.0f : -1.0f;
    }
}

// per-channel sum / sum of squares over M rows.  grid = (row chunks), block = 256.
// thread -> channel (tid % Cg) and row phase; fp64 accumulation; one partial row per workgroup, part[2][C][gridDim.x], which
// the BatchReNorm finalize folds in a fixed order (no floating-point atomics on the training path).
// gridDim.y > 1: micro-batch groups of M rows each (blockIdx.y = group), partial rows [group][gridDim.x] side by side.
__global__ __launch_bounds__(256) void moments_kernel(const float* x, int cs, int coff, long M, int C, double* part) {
    __shared__ double s1[256];
    __shared__ double s2[256];
    const int tid = threadIdx.x;
    x += (long)blockIdx.y * M * cs;
    const long prow = (long)blockIdx.y * gridDim.x + blockIdx.x, prows = (long)gridDim.x * gridDim.y;
    const int cpb = C < 256 ? C : 256;             // channels handled per pass
    const int rows_par = 256 / cpb;                // row phases
    for (int c0 = 0; c0 < C; c0 += cpb) {
        const int c = c0 + tid % cpb;
        const int rp = tid / cpb;
        double a = 0.0, b = 0.0;
        if (rp < rows_par && c < C) {
            for (long m = (long)blockIdx.x * rows_par + rp; m < M; m += (long)gridDim.x * rows_par) {
                const double v = (double)x[m * cs + coff + c];
                a += v;
                b += v * v;
            }
        }
        s1[tid] = a;
        s2[tid] = b;
        __syncthreads();
        if (tid < cpb && c0 + tid < C) {
            double ta = 0.0, tb = 0.0;
            for (int r = 0; r < rows_par; ++r) { ta += s1[r * cpb + tid]; tb += s2[r * cpb + tid]; }
            part[(long)(c0 + tid) * prows + prow] = ta;
            part[((long)C + c0 + tid) * prows + prow] = tb;
        }
        __syncthreads();
    }
}

}  // namespace dr
