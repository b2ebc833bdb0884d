// vote.h -- offset voting: per (sample, joint) top-5 pixels -> 3D candidates -> weighted mean-shift.
//
// Replaces the nested tf.map_fn graph of model/hourglass_um_crop_tiny.py:
//   _resume_om (:276-299), _xyz_estimation (:743-785), _generate_candidates (:598-627),
//   _get_candidate_weights (:629-682), _weighted_mean_shift (:684-741), then unnorm_xyz_pose
//   (data/preprocess.py:157-170) and the point cloud of generate_xyzs_from_multi_cfgs (:189-232).
//
// One workgroup = one sample x 4 joints; one wavefront per joint.  Phase 1 streams the maps once,
// coalesced, and parks the refined heat-map (hm+1)*hm3*[dm>=-0.99] of the 4 joints in LDS; phase 2
// is wave-local: five rounds of butterfly arg-max (ties -> lower pixel index, tf.nn.top_k), the
// 4x4x4 start cell (ties -> LAST cell, tf.where(...)[-1]) and ten mean-shift iterations.
// HBM-bound by construction: (5J+1)*npix*4 B read per crop (332 kB for J=16), 12J B written.
//
// Arithmetic is fp32 in the reference's op order with contraction off, and the one transcendental -- the mean-shift kernel weight
// exp(-d^2 / 2 sigma^2) -- is vote_exp below: a FIXED sequence of IEEE fp32 multiplies and adds (the Cephes / Eigen pexp<float>
// polynomial, which is what tf.exp runs on the reference's CPU path), restated operation by operation in oracle/pose.py::exp_f32.
// So the vote has NO deviation from the oracle: identical maps give bit-identical joints (tests/test_gpu_fullsize.py); a libm expf
// here differed from numpy's by an ulp, and ten mean-shift iterations between two candidate clusters amplified that ulp to 0.3 mm
// on a handful of knife-edge joints (rounds 1-5).  Documented choices: an out-of-range re-projected
// pixel contributes weight 0 (TF-GPU gather_nd), and a zero/non-finite kernel mass keeps the centre.
#pragma once
#include "dr_platform.h"
#include "kernels_misc.h"

namespace dr {

struct VoteParams {
    View hm, hm3, um;            // (B,h,w,J) (B,h,w,J) (B,h,w,3J)
    const float* tiny;           // (B,h,w) normalised depth at map resolution
    const float* cfg;            // (B,6) fx,fy,cx,cy,w,h of the crop camera
    const float* com;            // (B,3)
    float* xyz_mm;               // (B,3J) out, millimetres
    float* xyz_norm;             // nullable (B,3J) normalised
    int B, h, w, J;
};

constexpr int kVoteJC = 4;         // joints per workgroup (= waves)
constexpr int kVoteMaxPix = 4096;  // LDS: 4 joints x 4096 px x 4 B = 64 KiB

// exp(x) for x <= 0 (NaN propagates; x <= -87: 0, the result would be below the smallest normal).  Cody-Waite reduction x = n ln2 + r,
// degree-5 polynomial in r (Cephes expf / Eigen pexp<float> coefficients), scaling by 2^n through the exponent field; every step
// one correctly rounded fp32 operation, no fused multiply-add: bit-reproducible by any IEEE implementation.
__device__ __forceinline__ float vote_exp(float x) {
#pragma clang fp contract(off)
    if (!(x > -87.0f)) return x != x ? x : 0.0f;
    const float n = rintf(x * 1.44269504088896341f);
    float r = x - n * 0.693359375f;
    r = r - n * -2.12194440e-4f;
    float q = 1.9875691500e-4f;
    q = q * r + 1.3981999507e-3f;
    q = q * r + 8.3334519073e-3f;
    q = q * r + 4.1665795894e-2f;
    q = q * r + 1.6666665459e-1f;
    q = q * r + 5.0000001201e-1f;
    q = q * (r * r) + r;
    q = q + 1.0f;
    const int e = (int)n + 127;                                  // n >= -126: a normal power of two
#if defined(DR_EMU)
    float s; { const unsigned u = (unsigned)e << 23; memcpy(&s, &u, 4); }
#else
    const float s = __builtin_bit_cast(float, (unsigned)e << 23);
#endif
    return q * s;
}

__device__ __forceinline__ void vote_argmax_first(float& v, int& idx) {
    // wave-wide (max value, then min index)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float v2 = __shfl_xor(v, m);
        const int i2 = __shfl_xor(idx, m);
        if (v2 > v || (v2 == v && i2 < idx)) { v = v2; idx = i2; }
    }
}

__global__ __launch_bounds__(256) void vote_kernel(const VoteParams p) {
#pragma clang fp contract(off)
    DR_DYN_SMEM(smem_raw);
    float* ref = reinterpret_cast<float*>(smem_raw);      // [kVoteJC][npix]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int b = blockIdx.x;
    const int j0 = blockIdx.y * kVoteJC;
    const int npix = p.h * p.w;
    const int J = p.J;

    // ---- phase 1: refined heat-map of joints j0..j0+3 into LDS ---------------------------------
    for (int px = tid; px < npix; px += 256) {
        const long m = (long)b * npix + px;
        const float fg = (p.tiny[m] < -0.99f) ? 0.0f : 1.0f;
        const float* hm = p.hm.p + m * p.hm.cs + p.hm.coff;
        const float* h3 = p.hm3.p + m * p.hm3.cs + p.hm3.coff;
#pragma unroll
        for (int jj = 0; jj < kVoteJC; ++jj) {
            const int j = j0 + jj;
            float r = -INFINITY;
            if (j < J) {
                r = (hm[j] + 1.0f) * h3[j];
                r = r * fg;
            }
            ref[jj * npix + px] = r;
        }
    }
    __syncthreads();

    const int j = j0 + wave;
    if (j >= J) return;            // whole wave leaves together
    const float* rj = ref + wave * npix;

    // ---- top-5 (tf.nn.top_k sorted, ties -> lower index) ---------------------------------------
    int sel[5];
    unsigned long long taken = 0ull;           // bit i: my i-th pixel (lane + 64*i) already selected
    const int per_lane = (npix + 63) / 64;     // <= 64
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = 0; i < per_lane; ++i) {
            const int px = lane + 64 * i;
            if (px < npix && !((taken >> i) & 1ull)) {
                const float v = rj[px];
                if (v > bv || (v == bv && px < bi)) { bv = v; bi = px; }
            }
        }
        if (bi == 0x7fffffff) bv = -INFINITY;
        vote_argmax_first(bv, bi);
        if (bi == 0x7fffffff) bi = 0;          // degenerate (all NaN): fall back to pixel 0
        if ((bi & 63) == lane) taken |= 1ull << (bi >> 6);
        sel[k] = bi;
    }

    // ---- candidates + weights: lane i < 5 owns candidate i ------------------------------------
    const float* cfg = p.cfg + b * 6;
    const float cx0 = p.com[b * 3 + 0], cy0 = p.com[b * 3 + 1], cz0 = p.com[b * 3 + 2];
    const float w_ratio = cfg[4] / (float)p.w;
    const float h_ratio = cfg[5] / (float)p.h;
    const float fx = cfg[0] / w_ratio, fy = cfg[1] / h_ratio;
    const float cx = cfg[2] / w_ratio, cy = cfg[3] / h_ratio;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, wt = 0.f;
    {
        const int me = lane < 5 ? lane : 0;
        int px = sel[0];
#pragma unroll
        for (int k = 1; k < 5; ++k) px = (me == k) ? sel[k] : px;
        const long m = (long)b * npix + px;
        const float t = p.tiny[m];
        const float min_depth = cz0 - 300.0f * 0.5f;
        const float max_depth = cz0 + 300.0f * 0.5f;
        const float zz = (t < -0.99f) ? max_depth : (t * 300.0f + min_depth);
        const float xx = (float)(px % p.w), yy = (float)(px / p.w);
        float X = (xx - cx) * (zz / fx);
        float Y = (yy - cy) * (zz / fy);
        X = (X - cx0) / 100.0f;
        Y = (Y - cy0) / 100.0f;
        const float Z = (zz - cz0) / 100.0f;
        // resume_om: um * (0.8 - hm3*0.8)
        const float h3 = p.hm3.p[m * p.hm3.cs + p.hm3.coff + j];
        const float d3 = 0.8f - h3 * 0.8f;
        const float* um = p.um.p + m * p.um.cs + p.um.coff + 3 * j;
        c0 = X + um[0] * d3;
        c1 = Y + um[1] * d3;
        c2 = Z + um[2] * d3;
        // weight = hm at the re-projected pixel (raw hm; out of range -> 0)
        const float ux = c0 * 100.0f + cx0, uy = c1 * 100.0f + cy0, uz = c2 * 100.0f + cz0;
        const float uf = (ux * fx / uz + cx) + 0.5f;
        const float vf = (uy * fy / uz + cy) + 0.5f;
        wt = 0.f;
        if (uf > -1.0f && uf < (float)p.w && vf > -1.0f && vf < (float)p.h) {      // NaN fails
            const int uu = (int)uf, vv = (int)vf;                                      // trunc toward 0
            const long mm = (long)b * npix + vv * p.w + uu;
            wt = p.hm.p[mm * p.hm.cs + p.hm.coff + j];
        }
    }
    float cp[5][3], cw[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        cp[i][0] = __shfl(c0, i);
        cp[i][1] = __shfl(c1, i);
        cp[i][2] = __shfl(c2, i);
        cw[i] = __shfl(wt, i);
    }

    // ---- start cell: 4x4x4 weight histogram, lane = cell; max, ties -> last cell ---------------
    float cell_w = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        int q[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float f = (cp[i][d] + 1.0f) * 2.0f;
            f = fminf(fmaxf(f, 0.0f), 3.9f);
            q[d] = (int)f;                      // NaN -> 0 via fmaxf/fminf semantics
        }
        const int cell = q[0] * 16 + q[1] * 4 + q[2];
        if (cell == lane) cell_w = cell_w + cw[i];
    }
    float bv = cell_w;
    int bi = lane;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float v2 = __shfl_xor(bv, m);
        const int i2 = __shfl_xor(bi, m);
        if (v2 > bv || (v2 == bv && i2 > bi)) { bv = v2; bi = i2; }
    }
    float ctr[3];
    ctr[0] = (float)(bi >> 4) / 2.0f - 1.0f + 0.25f;
    ctr[1] = (float)((bi >> 2) & 3) / 2.0f - 1.0f + 0.25f;
    ctr[2] = (float)(bi & 3) / 2.0f - 1.0f + 0.25f;

    // ---- weighted mean-shift, 10 iterations, bandwidth 0.4 -------------------------------------
    const float inv_sigma = -3.125f;   // -1/(2*0.4^2) evaluated in double by the reference (:738), then cast
#pragma unroll 1
    for (int it = 0; it < 10; ++it) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, ss = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const float d0 = cp[i][0] - ctr[0], d1 = cp[i][1] - ctr[1], d2 = cp[i][2] - ctr[2];
            float s = (d0 * d0 + d1 * d1) + d2 * d2;
            s = vote_exp(inv_sigma * s) * cw[i];
            a0 = a0 + cp[i][0] * s;
            a1 = a1 + cp[i][1] * s;
            a2 = a2 + cp[i][2] * s;
            ss = ss + s;
        }
        if (ss == 0.0f || !(fabsf(ss) < INFINITY)) break;
        ctr[0] = a0 / ss;
        ctr[1] = a1 / ss;
        ctr[2] = a2 / ss;
    }
    if (lane < 3) {
        const float cn = lane == 0 ? ctr[0] : (lane == 1 ? ctr[1] : ctr[2]);
        const float cm = lane == 0 ? cx0 : (lane == 1 ? cy0 : cz0);
        const long o = (long)b * 3 * J + 3 * j + lane;
        if (p.xyz_norm) p.xyz_norm[o] = cn;
        p.xyz_mm[o] = cn * 100.0f + cm;
    }
}

}  // namespace dr
