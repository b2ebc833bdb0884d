"""numpy restatement of the pre/post-processing on the hot path (TEST INFRASTRUCTURE).

Parity unpinned (see oracle/__init__.py).  Everything is float32 op-by-op in the
order the reference builds its TF graph, so that the HIP kernels can be compared
bit-closely; index math follows TF cast semantics (truncate toward zero).

* ``norm_dm``                 data/preprocess.py:176-187
* ``generate_xyzs``           data/preprocess.py:189-232
* ``norm_xyz_pose`` / ``unnorm_xyz_pose``   data/preprocess.py:144-170
* ``hm_2d`` / ``hm_3d`` / ``um_gt``          model/hourglass_um_crop_tiny.py:193-274
* ``resume_om``               model/hourglass_um_crop_tiny.py:276-299
* ``xyz_estimation``          model/hourglass_um_crop_tiny.py:598-785 (the vote)
* ``mean_jnt_error`` / ``max_jnt_error``     data/evaluation.py:9-18
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
D_RANGE = f32(300.0)            # preprocess.py:172
POSE_NORM_RATIO = f32(100.0)    # preprocess.py:173
MAX_DIST_2D = f32(4.0)          # hourglass_um_crop_tiny.py:193
MAX_DIST_3D = f32(0.8)          # hourglass_um_crop_tiny.py:194


def norm_dm(dms: np.ndarray, coms: np.ndarray) -> np.ndarray:
    dms = dms.astype(f32)
    out = np.empty_like(dms)
    for b in range(dms.shape[0]):
        max_depth = f32(coms[b, 2]) + D_RANGE * f32(0.5)
        min_depth = f32(coms[b, 2]) - D_RANGE * f32(0.5)
        dm = dms[b]
        mask = (dm < max_depth) & (dm > (min_depth - D_RANGE * f32(0.5)))
        out[b] = np.where(mask, (dm - min_depth) / D_RANGE, f32(-1.0))
    return out


def _scaled_cfg(cfg, w, h):
    """CameraConfig(cfg/ratio) used at hm_2d (:225-229), xyzs (preprocess.py:213-217), weights (:649-653)."""
    w_ratio = f32(cfg[4]) / f32(w)
    h_ratio = f32(cfg[5]) / f32(h)
    return (f32(cfg[0]) / w_ratio, f32(cfg[1]) / h_ratio, f32(cfg[2]) / w_ratio, f32(cfg[3]) / h_ratio)


def generate_xyzs(dms: np.ndarray, cfgs: np.ndarray, coms: np.ndarray) -> np.ndarray:
    """dms (B,h,w,1) normalised -> (B,h,w,3) normalised point cloud."""
    B, h, w, _ = dms.shape
    out = np.empty((B, h, w, 3), f32)
    xx, yy = np.meshgrid(np.arange(h), np.arange(w))      # 'xy': xx[i,j]=j, yy[i,j]=i
    xx, yy = xx.astype(f32), yy.astype(f32)
    for b in range(B):
        zz = dms[b, :, :, 0].astype(f32)
        com = coms[b].astype(f32)
        min_depth = com[2] - D_RANGE * f32(0.5)
        max_depth = com[2] + D_RANGE * f32(0.5)
        zz = np.where(zz < f32(-0.99), max_depth, zz * D_RANGE + min_depth).astype(f32)
        fx, fy, cx, cy = _scaled_cfg(cfgs[b], w, h)
        x = (xx - cx) * (zz / fx)
        y = (yy - cy) * (zz / fy)
        out[b, :, :, 0] = (x - com[0]) / POSE_NORM_RATIO
        out[b, :, :, 1] = (y - com[1]) / POSE_NORM_RATIO
        out[b, :, :, 2] = (zz - com[2]) / POSE_NORM_RATIO
    return out


def norm_xyz_pose(poses: np.ndarray, coms: np.ndarray) -> np.ndarray:
    J = poses.shape[1] // 3
    return ((poses.astype(f32) - np.tile(coms.astype(f32), (1, J))) / POSE_NORM_RATIO).astype(f32)


def unnorm_xyz_pose(nposes: np.ndarray, coms: np.ndarray) -> np.ndarray:
    J = nposes.shape[1] // 3
    return (nposes.astype(f32) * POSE_NORM_RATIO + np.tile(coms.astype(f32), (1, J))).astype(f32)


def xyz2uvd(xyz: np.ndarray, cfg4) -> np.ndarray:
    """data/util.py:20 _pro: u = x*fx/z + cx."""
    fx, fy, cx, cy = cfg4
    xyz = xyz.reshape(-1, 3).astype(f32)
    u = xyz[:, 0] * fx / xyz[:, 2] + cx
    v = xyz[:, 1] * fy / xyz[:, 2] + cy
    return np.stack([u, v, xyz[:, 2]], axis=1).astype(f32)


def hm_2d(poses: np.ndarray, cfgs: np.ndarray, out_h=32, out_w=32) -> np.ndarray:
    B, J = poses.shape[0], poses.shape[1] // 3
    xx, yy = np.meshgrid(np.arange(out_h), np.arange(out_w))
    xx, yy = xx.astype(f32)[:, :, None], yy.astype(f32)[:, :, None]
    out = np.empty((B, out_h, out_w, J), f32)
    for b in range(B):
        uvd = xyz2uvd(poses[b], _scaled_cfg(cfgs[b], out_w, out_h))
        uu, vv = uvd[:, 0].reshape(1, 1, J), uvd[:, 1].reshape(1, 1, J)
        d = np.sqrt(np.square(xx - uu) + np.square(yy - vv))
        out[b] = np.maximum(MAX_DIST_2D - d, f32(0)) / MAX_DIST_2D
    return out


def hm_3d(oms: np.ndarray) -> np.ndarray:
    B, h, w, J3 = oms.shape
    o = oms.reshape(B, h, w, J3 // 3, 3).astype(f32)
    d = np.sqrt(o[..., 0] * o[..., 0] + o[..., 1] * o[..., 1] + o[..., 2] * o[..., 2])
    return np.maximum((MAX_DIST_3D - d) / MAX_DIST_3D, f32(0)).astype(f32)


def um_gt(oms: np.ndarray, hm3: np.ndarray) -> np.ndarray:
    B, h, w, J3 = oms.shape
    o = oms.reshape(B, h, w, J3 // 3, 3).astype(f32)
    dm3 = (MAX_DIST_3D - hm3 * MAX_DIST_3D).astype(f32)
    mask = dm3 < f32(MAX_DIST_3D - f32(1e-2))
    with np.errstate(divide='ignore', invalid='ignore'):
        um = np.where(mask[..., None], o / dm3[..., None], f32(0))
    return um.reshape(B, h, w, J3).astype(f32)


def make_targets(dms_norm: np.ndarray, poses: np.ndarray, cfgs: np.ndarray, coms: np.ndarray, out_hw=32):
    """hourglass_um_crop_tiny.py:336-346 (without augmentation). Returns gt_hm, gt_hm3, gt_um."""
    s = dms_norm.shape[1] // out_hw
    gt_hms = hm_2d(poses, cfgs, out_hw, out_hw)
    npose = norm_xyz_pose(poses, coms)
    tiny = dms_norm[:, ::s, ::s, :]
    xyzs = generate_xyzs(tiny, cfgs, coms)
    J = poses.shape[1] // 3
    gt_oms = (npose.reshape(-1, 1, 1, 3 * J) - np.tile(xyzs, (1, 1, 1, J))).astype(f32)
    gt_hm3 = hm_3d(gt_oms)
    gt_um = um_gt(gt_oms, gt_hm3)
    return gt_hms, gt_hm3, gt_um


def resume_om(hm3: np.ndarray, um: np.ndarray) -> np.ndarray:
    B, h, w, J = hm3.shape
    dm3 = (MAX_DIST_3D - hm3.astype(f32) * MAX_DIST_3D).astype(f32)
    return (um.reshape(B, h, w, J, 3).astype(f32) * dm3[..., None]).reshape(B, h, w, 3 * J).astype(f32)


# --------------------------------------------------------------------------------------
# The vote.
# --------------------------------------------------------------------------------------
NUM_PT = 5
MS_ITERS = 10
MS_BANDWIDTH = 0.4          # python float in the reference (:775); -1/(2*bw*bw) -> f32(-3.125)


def top_k_indices(v: np.ndarray, k: int) -> np.ndarray:
    """tf.nn.top_k(sorted=True): descending value, ties -> lower index first."""
    order = np.lexsort((np.arange(v.size), -v.astype(np.float64)))
    return order[:k]


def candidate_weight(p_norm, com, cfg4, hm_j, out_hw=32):
    """hourglass_um_crop_tiny.py:646-664 for ONE candidate.  gather_nd out-of-range => weight 0
    (TF-GPU behaviour)."""
    p = p_norm.astype(f32) * POSE_NORM_RATIO + com.astype(f32)
    fx, fy, cx, cy = cfg4
    with np.errstate(all='ignore'):
        u = p[0] * fx / p[2] + cx
        v = p[1] * fy / p[2] + cy
        uf, vf = f32(u + f32(0.5)), f32(v + f32(0.5))
    if not (np.isfinite(uf) and np.isfinite(vf)):
        return f32(0)
    if not (uf > -1 and uf < out_hw and vf > -1 and vf < out_hw):
        return f32(0)
    uu, vv = int(uf), int(vf)          # trunc toward zero; (-1,0) -> 0
    return f32(hm_j[vv, uu])


def exp_f32(x) -> np.float32:
    """exp(x) for x <= 0 as a FIXED sequence of IEEE fp32 operations: Cody-Waite reduction, the degree-5 polynomial of Cephes expf --
    the algorithm (and coefficients) of Eigen's pexp<float>, which is the kernel behind tf.exp on the reference's CPU path
    (hourglass_um_crop_tiny.py:715-721 builds the mean-shift weights with tf.exp) -- every step rounded to fp32, no fused
    multiply-add.  The HIP vote computes the same sequence (densereg_amd/csrc/vote.h::vote_exp), so the vote is bit-reproducible
    between the two; against a correctly rounded exp the result is within 1 ulp.  x <= -87 gives 0, NaN propagates."""
    x = f32(x)
    if not (x > f32(-87.0)):
        return x if x != x else f32(0)
    n = f32(np.rint(f32(x * f32(1.44269504088896341))))
    r = f32(x - f32(n * f32(0.693359375)))
    r = f32(r - f32(n * f32(-2.12194440e-4)))
    q = f32(1.9875691500e-4)
    for c in (1.3981999507e-3, 8.3334519073e-3, 4.1665795894e-2, 1.6666665459e-1, 5.0000001201e-1):
        q = f32(f32(q * r) + f32(c))
    q = f32(f32(q * f32(r * r)) + r)
    q = f32(q + f32(1.0))
    s = np.array((int(n) + 127) << 23, np.uint32).view(np.float32)
    return f32(q * s)


def weighted_mean_shift(can: np.ndarray, w: np.ndarray, num_it=MS_ITERS, band_width=MS_BANDWIDTH):
    """hourglass_um_crop_tiny.py:694-724 for one joint. can (n,3), w (n,)."""
    can = can.astype(f32)
    w = w.astype(f32)
    num_quan = f32(2.0)
    q = np.clip((can + f32(1.0)) * num_quan, f32(0), f32(2 * 2.0 - 0.1)).astype(np.int64)
    grid = np.zeros((4, 4, 4), f32)
    for i in range(can.shape[0]):
        grid[q[i, 0], q[i, 1], q[i, 2]] += w[i]
    flat = grid.reshape(-1)
    last = int(np.nonzero(flat == flat.max())[0][-1])          # tf.where(...)[-1], row-major
    idx = np.array([last // 16, (last // 4) % 4, last % 4], f32)
    c = idx / num_quan - f32(1.0) + f32(0.5) / num_quan
    inv_sigma = f32(-1.0 / (2 * float(band_width) * float(band_width)))
    for _ in range(num_it):
        acc = np.zeros(3, f32)
        ssum = f32(0)
        for i in range(can.shape[0]):
            d0, d1, d2 = can[i, 0] - c[0], can[i, 1] - c[1], can[i, 2] - c[2]
            s = f32(d0 * d0 + d1 * d1) + d2 * d2
            s = exp_f32(f32(inv_sigma * s)) * w[i]
            acc = (acc + can[i] * s).astype(f32)
            ssum = f32(ssum + s)
        if ssum == 0 or not np.isfinite(ssum):
            break                       # guard the reference lacks
        c = (acc / ssum).astype(f32)
    return c


def xyz_estimation(hms, oms, hm3s, dms32, cfgs, coms, return_debug=False):
    """The vote: (B,h,w,J),(B,h,w,3J),(B,h,w,J),(B,h,w,1),(B,6),(B,3) -> (B,J,3) normalised."""
    B, h, w, J = hms.shape
    hms, oms, hm3s, dms32 = (a.astype(f32) for a in (hms, oms, hm3s, dms32))
    xyzs = np.tile(generate_xyzs(dms32, cfgs, coms), (1, 1, 1, J)) + oms
    refined = (hms + f32(1.0)) * hm3s
    refined = refined * np.where(dms32 < f32(-0.99), f32(0), f32(1))
    out = np.zeros((B, J, 3), f32)
    dbg = {'idx': np.zeros((B, J, NUM_PT), np.int64), 'can': np.zeros((B, J, NUM_PT, 3), f32),
           'w': np.zeros((B, J, NUM_PT), f32)}
    for b in range(B):
        cfg4 = _scaled_cfg(cfgs[b], w, h)
        ref_b = refined[b].reshape(-1, J)
        xyz_b = xyzs[b].reshape(-1, 3 * J)
        for j in range(J):
            idx = top_k_indices(ref_b[:, j], NUM_PT)
            can = xyz_b[idx, 3 * j:3 * j + 3]
            wts = np.array([candidate_weight(can[i], coms[b], cfg4, hms[b, :, :, j], h)
                            for i in range(NUM_PT)], f32)
            out[b, j] = weighted_mean_shift(can, wts)
            dbg['idx'][b, j], dbg['can'][b, j], dbg['w'][b, j] = idx, can, wts
    return (out, dbg) if return_debug else out


def estimate_pose_mm(hm, hm3, um, dm_norm, cfgs, coms, out_hw=32):
    """JointDetectionModel.test (:451-462): last-stack maps -> (B,3J) xyz in mm."""
    s = dm_norm.shape[1] // out_hw
    tiny = dm_norm[:, ::s, ::s, :]
    om = resume_om(hm3, um)
    n = xyz_estimation(hm, om, hm3, tiny, cfgs, coms)
    return unnorm_xyz_pose(n.reshape(n.shape[0], -1), coms)


def mean_jnt_error(a, b):
    return float(np.linalg.norm(a.reshape(-1, 3) - b.reshape(-1, 3), axis=1).mean())


def max_jnt_error(a, b):
    return float(np.linalg.norm(a.reshape(-1, 3) - b.reshape(-1, 3), axis=1).max())
