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
