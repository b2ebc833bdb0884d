"""Seeded synthetic depth crops.

The reference's datasets (ICVL/NYU/MSRA TFRecords, ``data/dataset_base.py``) are not
available; BASELINE.json quotes its metric on synthetic 128x128 crops.  A sample is what
``dataset.preprocess_op`` hands to the model (``hourglass_um_crop_tiny.py:133-141``):
``dm`` (128,128,1) depth in mm (0 = background), ``pose`` (3J,) xyz in mm, ``cfg``
(fx,fy,cx,cy,w,h) of the *cropped* camera, ``com`` (3,) centre of mass.
"""
from __future__ import annotations

import numpy as np

# Camera intrinsics and joint counts of the three datasets
# (data/icvl.py:12-17, data/nyu.py:13,40-45, data/msra.py:13-17).
DATASETS = {
    'icvl': dict(fx=241.42, fy=241.42, cx=160.0, cy=120.0, w=320, h=240, jnt_num=16, exact_num=1596,
                 approximate_num=220 * 101, epochs_per_decay=10),   # icvl decay undefined in ref (Appendix C.1)
    'nyu': dict(fx=588.235, fy=587.084, cx=320.0, cy=240.0, w=640, h=480, jnt_num=14, exact_num=8252,
                approximate_num=730 * 101, epochs_per_decay=10),
    'msra': dict(fx=241.42, fy=241.42, cx=160.0, cy=120.0, w=320, h=240, jnt_num=21, exact_num=8499,
                 approximate_num=85 * 801, epochs_per_decay=20),
}


def center_of_mass(dm: np.ndarray, cfg: np.ndarray) -> np.ndarray:
    """data/preprocess.py:131-142."""
    h, w = dm.shape[:2]
    fg = dm[dm > 0]
    ave_d = float(fg.mean()) if fg.size else 0.0
    ave_d = max(ave_d, 200.0)
    ave_x = (w / 2 - cfg[2]) * ave_d / cfg[0]
    ave_y = (h / 2 - cfg[3]) * ave_d / cfg[1]
    return np.array([ave_x, ave_y, ave_d], np.float32)


def make_crops(batch: int, dataset: str = 'icvl', seed: int = 20240, rank: int = 0, hw: int = 128):
    """Returns dm (B,hw,hw,1) mm, pose (B,3J) mm, cfg (B,6), com (B,3), names list."""
    ds = DATASETS[dataset]
    J = ds['jnt_num']
    rng = np.random.Generator(np.random.PCG64(seed + rank))
    dms = np.zeros((batch, hw, hw, 1), np.float32)
    poses = np.zeros((batch, 3 * J), np.float32)
    cfgs = np.zeros((batch, 6), np.float32)
    coms = np.zeros((batch, 3), np.float32)
    vv, uu = np.meshgrid(np.arange(hw, dtype=np.float32), np.arange(hw, dtype=np.float32), indexing='ij')
    c = hw / 2.0
    sc = hw / 128.0
    for b in range(batch):
        com_z = rng.uniform(250.0, 900.0)
        L = rng.uniform(80.0, 160.0)
        cfg = np.array([ds['fx'] * hw / L, ds['fy'] * hw / L, c + rng.normal(0, 3), c + rng.normal(0, 3), hw, hw],
                       np.float32)
        # silhouette: palm disc + five radial "finger" capsules
        r = np.hypot(uu - c, vv - c)
        fg = r < 44.0 * sc
        base = rng.uniform(0, 2 * np.pi)
        for f in range(5):
            ang = base + f * (2 * np.pi / 5) + rng.normal(0, 0.15)
            du, dv = np.cos(ang), np.sin(ang)
            t = np.clip((uu - c) * du + (vv - c) * dv, 0, 62.0 * sc)
            dist = np.hypot(uu - c - t * du, vv - c - t * dv)
            fg |= dist < 5.0 * sc
        depth = com_z + 60.0 * np.sin(uu / (9.0 * sc)) * np.cos(vv / (11.0 * sc)) + rng.normal(0, 4.0, (hw, hw))
        depth = np.clip(depth, com_z - 140.0, com_z + 140.0)
        dm = np.where(fg, depth, 0.0).astype(np.float32)
        com = center_of_mass(dm, cfg)
        ys, xs = np.nonzero(fg)
        pick = rng.choice(ys.size, J, replace=False)
        pose = np.zeros((J, 3), np.float32)
        for j, k in enumerate(pick):
            d = dm[ys[k], xs[k]]
            pose[j] = [(xs[k] - cfg[2]) * d / cfg[0], (ys[k] - cfg[3]) * d / cfg[1], d]   # data/util.py:21
        # joints sit a few mm off the surface (a joint exactly on a map pixel makes the reference's
        # unit-offset target 0/0, hourglass_um_crop_tiny.py:268-272)
        pose += rng.uniform(1.0, 6.0, (J, 3)).astype(np.float32) * rng.choice([-1.0, 1.0], (J, 3)).astype(np.float32)
        dms[b, :, :, 0], poses[b], cfgs[b], coms[b] = dm, pose.reshape(-1), cfg, com
    names = ['synthetic_%s/rank%d_%07d.png' % (dataset, rank, i) for i in range(batch)]
    return dms, poses, cfgs, coms, names


def make_hand_crops(batch: int, dataset: str = 'icvl', seed: int = 7, rank: int = 0, hw: int = 128):
    """LEARNABLE synthetic crops: an articulated "hand" whose joints are a function of what the camera sees -- joint 0 the palm
    centre, the others at equal fractions along five fingers in a fixed cyclic order (finger f = (j-1) % 5), every joint 6 mm
    behind the visible surface.  ``make_crops`` (the throughput workload) picks its joints at random among the foreground pixels:
    nothing there to learn.  Same return convention; used by the training tests and ``examples/train_synthetic.py``, where the
    engine's own training has to make the heat-maps peaked (hourglass_um_crop_tiny.py:193-274 targets, :323-371 loss)."""
    ds = DATASETS[dataset]
    J = ds['jnt_num']
    rng = np.random.Generator(np.random.PCG64(1_000_003 * seed + rank))
    dms = np.zeros((batch, hw, hw, 1), np.float32)
    poses = np.zeros((batch, 3 * J), np.float32)
    cfgs = np.zeros((batch, 6), np.float32)
    coms = np.zeros((batch, 3), np.float32)
    vv, uu = np.meshgrid(np.arange(hw, dtype=np.float32), np.arange(hw, dtype=np.float32), indexing='ij')
    sc = hw / 128.0
    nseg = -(-(J - 1) // 5)                                   # joints per finger (the last fingers may have one fewer)
    for b in range(batch):
        com_z = rng.uniform(300.0, 800.0)
        L = rng.uniform(100.0, 140.0)
        cu, cv = hw / 2.0 + rng.normal(0, 4.0 * sc), hw / 2.0 + rng.normal(0, 4.0 * sc)      # where the palm sits in the crop
        cfg = np.array([ds['fx'] * hw / L, ds['fy'] * hw / L, hw / 2.0 + rng.normal(0, 2), hw / 2.0 + rng.normal(0, 2), hw, hw], np.float32)
        rot = rng.normal(0, 0.12)
        palm_r = rng.uniform(20.0, 24.0) * sc
        tiltu, tiltv = rng.normal(0, 0.35, 2)                 # the palm plane's slope (mm of depth per pixel)
        depth = com_z + tiltu * (uu - cu) + tiltv * (vv - cv) - 12.0 * np.exp(-((uu - cu) ** 2 + (vv - cv) ** 2) / (2 * (palm_r * 0.8) ** 2))
        fg = np.hypot(uu - cu, vv - cv) < palm_r
        tips = []
        for f in range(5):
            ang = rot - 0.5 * np.pi + (f - 2) * 0.52 + rng.normal(0, 0.07)       # a fan of five fingers, 30 degrees apart, pointing up
            du, dv = np.cos(ang), np.sin(ang)
            length = rng.uniform(44.0, 56.0) * sc * (0.8 if f in (0, 4) else 1.0)
            curl = rng.normal(0, 0.5)                         # depth slope along the finger (mm per pixel): towards / away from the camera
            t = np.clip((uu - cu) * du + (vv - cv) * dv, 0, length)
            dist = np.hypot(uu - cu - t * du, vv - cv - t * dv)
            finger = (dist < 4.5 * sc) & ~fg
            zf = com_z + tiltu * (t * du) + tiltv * (t * dv) + curl * np.maximum(t - palm_r, 0.0) - 3.0 * np.cos(np.clip(dist / (4.5 * sc), 0, 1) * np.pi / 2)
            depth = np.where(finger, zf, depth)
            fg |= finger
            tips.append((du, dv, length, curl))
        depth = depth + rng.normal(0, 1.0, (hw, hw))
        dm = np.where(fg, np.clip(depth, com_z - 140.0, com_z + 140.0), 0.0).astype(np.float32)
        com = center_of_mass(dm, cfg)
        pose = np.zeros((J, 3), np.float32)

        def lift(u, v):
            iu, iv = int(np.clip(round(u), 0, hw - 1)), int(np.clip(round(v), 0, hw - 1))
            d = dm[iv, iu] if dm[iv, iu] > 0 else com_z
            d = d + 6.0                                       # inside the hand, behind the surface the camera sees
            return [(u - cfg[2]) * d / cfg[0], (v - cfg[3]) * d / cfg[1], d]          # data/util.py:21
        pose[0] = lift(cu, cv)
        for j in range(1, J):
            f, sgm = (j - 1) % 5, (j - 1) // 5
            du, dv, length, _ = tips[f]
            t = palm_r + (length - palm_r - 2.0 * sc) * (sgm + 1) / nseg
            pose[j] = lift(cu + t * du, cv + t * dv)
        dms[b, :, :, 0], poses[b], cfgs[b], coms[b] = dm, pose.reshape(-1), cfg, com
    names = ['synthetic_hand_%s/rank%d_%07d.png' % (dataset, rank, i) for i in range(batch)]
    return dms, poses, cfgs, coms, names
