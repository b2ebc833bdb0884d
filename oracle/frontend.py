"""CPU oracle of the crop + centre-of-mass front-end and the training-time augmentation.

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (same status as the rest of ``oracle/``: the TF-1.3 / Python-2
reference can be neither imported nor built here, and it ships no tests or golden vectors for these
functions).  Plain numpy restatement, float32 arithmetic in the reference's op order, of

* ``crop_from_xyz_pose``      /root/reference data/preprocess.py:10-79
* ``crop_from_bbx``           data/preprocess.py:81-129
* ``center_of_mass``          data/preprocess.py:131-142
* ``data_aug``                data/preprocess.py:234-268
* ``_pro`` / ``_bpro``        data/util.py:20-21

TensorFlow kernels the reference calls and how they are restated  [TF1.3-semantics, unverifiable here]:

* ``tf.cast(float -> int32)`` truncates toward zero.
* ``tf.divide(int, int)`` is true division (float64), ``tf.to_int32`` truncates it again.
* ``tf.image.resize_images(.., (h, w))`` = bilinear, ``align_corners=False``: ``scale = in/out`` (float32),
  ``src = dst*scale``, ``lo = int(src)``, ``hi = min(ceil(src), in-1)``, ``lerp = src - lo``;
  ``top = tl + (tr-tl)*xl``, ``bot = bl + (br-bl)*xl``, ``out = top + (bot-top)*yl``.
* ``tf.image.resize_images(.., method=1)`` = nearest: ``src = min(int(floor(dst*scale)), in-1)``.
* ``tf.image.resize_image_with_crop_or_pad``: centre crop offset ``max(-diff // 2, 0)``, pad offset
  ``max(diff // 2, 0)`` with ``diff = target - size`` (floor division).
* ``tf.contrib.image.rotate(img, a)`` = projective transform, nearest, zero fill: output pixel (x, y) reads input
  ``(round(c*x - s*y + ox), round(s*x + c*y + oy))`` with ``ox = ((W-1) - (c*(W-1) - s*(H-1)))/2``,
  ``oy = ((H-1) - (s*(W-1) + c*(H-1)))/2`` and ``round`` = half away from zero.
* ``tf.reduce_min`` of an empty tensor is +max-float: the NYU/MSRA depth threshold then keeps everything;
  ``tf.reduce_mean`` of an empty mask is NaN and ``tf.maximum(NaN, 200)`` is NaN in TF -- an all-background
  crop has no defined centre of mass in the reference; this restatement (and the engine) return depth 200.
"""
import numpy as np

F32 = np.float32


def pro(xyz, cfg):
    """data/util.py:20 -- perspective projection of (J,3) points with cfg = [fx, fy, cx, cy, w, h]."""
    xyz = np.asarray(xyz, F32).reshape(-1, 3)
    cfg = np.asarray(cfg, F32)
    u = xyz[:, 0] * cfg[0] / xyz[:, 2] + cfg[2]
    v = xyz[:, 1] * cfg[1] / xyz[:, 2] + cfg[3]
    return np.stack([u, v, xyz[:, 2]], 1).astype(F32)


def bpro(uvd, cfg):
    """data/util.py:21 -- back projection."""
    uvd = np.asarray(uvd, F32).reshape(-1, 3)
    cfg = np.asarray(cfg, F32)
    x = (uvd[:, 0] - cfg[2]) * uvd[:, 2] / cfg[0]
    y = (uvd[:, 1] - cfg[3]) * uvd[:, 2] / cfg[1]
    return np.stack([x, y, uvd[:, 2]], 1).astype(F32)


def _box_from_pose(pose, cfg, pad):
    """preprocess.py:25-38 -- integer crop box (top, left, bottom, right) around the projected joints."""
    uvd = pro(pose, cfg)
    mn, mx = uvd.min(0), uvd.max(0)
    pad = F32(pad)
    h, w = F32(cfg[5]), F32(cfg[4])
    top = min(max(mn[1] - pad, F32(0)), h - 2 * pad)
    left = min(max(mn[0] - pad, F32(0)), w - 2 * pad)
    bottom = max(min(mx[1] + pad, h), F32(top) + 2 * pad - 1)
    right = max(min(mx[0] + pad, w), F32(left) + 2 * pad - 1)
    return int(top), int(left), int(bottom), int(right), uvd


def resize_bilinear(img, out_h, out_w):
    """tf.image.resize_images default (bilinear, align_corners=False, legacy coordinates), float32."""
    img = np.asarray(img, F32)
    in_h, in_w = img.shape
    out = np.empty((out_h, out_w), F32)
    sy, sx = F32(in_h) / F32(out_h), F32(in_w) / F32(out_w)
    xs = (np.arange(out_w, dtype=F32) * sx).astype(F32)
    x0 = xs.astype(np.int64)
    x1 = np.minimum(np.ceil(xs).astype(np.int64), in_w - 1)
    xl = (xs - x0.astype(F32)).astype(F32)
    for oy in range(out_h):
        ys = F32(oy) * sy
        y0 = int(ys)
        y1 = min(int(np.ceil(ys)), in_h - 1)
        yl = F32(ys - F32(y0))
        tl, tr, bl, br = img[y0, x0], img[y0, x1], img[y1, x0], img[y1, x1]
        top = (tl + (tr - tl) * xl).astype(F32)
        bot = (bl + (br - bl) * xl).astype(F32)
        out[oy] = top + (bot - top) * yl
    return out


def _crop_pad_resize(dm, top, left, bottom, right, out_h, out_w):
    """crop_to_bounding_box + pad_to_bounding_box (centred in a square) + bilinear resize, preprocess.py:40-53."""
    crop = np.asarray(dm, F32)[top:bottom, left:right]
    h, w = bottom - top, right - left
    longer = max(h, w)
    off_h = int((longer - h) / 2)
    off_w = int((longer - w) / 2)
    sq = np.zeros((longer, longer), F32)
    sq[off_h:off_h + h, off_w:off_w + w] = crop
    return resize_bilinear(sq, out_h, out_w), longer, off_h, off_w


def _new_cfg(cfg, top, left, longer, off_h, off_w, out_h, out_w):
    """preprocess.py:69-78."""
    cfg = np.asarray(cfg, F32)
    rx, ry = F32(longer / out_w), F32(longer / out_h)
    return np.array([cfg[0] / rx, cfg[1] / ry, (cfg[2] - F32(left) + F32(off_w)) / rx, (cfg[3] - F32(top) + F32(off_h)) / ry,
                     F32(out_w), F32(out_h)], F32)


def crop_from_xyz_pose(dm, pose, cfg, out_w, out_h, pad=20.0, dataset='nyu'):
    """preprocess.py:10-79.  Returns (crop, pose, new_cfg)."""
    dm = np.asarray(dm, F32)
    in_h, in_w = dm.shape
    top, left, bottom, right, uvd = _box_from_pose(pose, cfg, pad)
    crop, longer, off_h, off_w = _crop_pad_resize(dm, top, left, bottom, right, out_h, out_w)
    uu = np.clip(uvd[:, 0].astype(np.int32), 0, in_w - 1)
    vv = np.clip(uvd[:, 1].astype(np.int32), 0, in_h - 1)
    dd = dm[vv, uu]
    dd = dd[dd > 100]
    d_th = F32(dd.min() + F32(250.0)) if dd.size else F32(np.finfo(np.float32).max)
    if dataset == 'icvl':
        d_th = F32(500.0)
    crop = np.where(crop < d_th, crop, F32(0)).astype(F32)
    return crop, np.asarray(pose, F32), _new_cfg(cfg, top, left, longer, off_h, off_w, out_h, out_w)


def crop_from_bbx(dm, pose, bbx, cfg, out_w, out_h):
    """preprocess.py:81-129; bbx = [top, left, bottom, right, d_th]."""
    top, left, bottom, right = (int(F32(v)) for v in bbx[:4])
    crop, longer, off_h, off_w = _crop_pad_resize(dm, top, left, bottom, right, out_h, out_w)
    crop = np.where(crop < F32(bbx[4]), crop, F32(0)).astype(F32)
    return crop, np.asarray(pose, F32), _new_cfg(cfg, top, left, longer, off_h, off_w, out_h, out_w)


def center_of_mass(dm, cfg):
    """preprocess.py:131-142: mean depth of the positive pixels at the centre pixel of the crop."""
    dm = np.asarray(dm, F32)
    cfg = np.asarray(cfg, F32)
    c_h, c_w = dm.shape
    ave_u, ave_v = F32(c_w / 2), F32(c_h / 2)
    pos = dm[dm > 0]
    # float32 mean accumulated in float64 (Eigen's tree reduction is within 1 ulp of this for 16384 values)
    ave_d = F32(pos.astype(np.float64).mean()) if pos.size else F32(200.0)
    ave_d = max(ave_d, F32(200.0))
    return np.array([(ave_u - cfg[2]) * ave_d / cfg[0], (ave_v - cfg[3]) * ave_d / cfg[1], ave_d], F32)


def _round_half_away(x):
    return np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5)).astype(np.int64)


def rotate_nearest(img, angle):
    """tf.contrib.image.rotate (nearest, zero fill)."""
    img = np.asarray(img, F32)
    H, W = img.shape
    c, s = F32(np.cos(F32(angle))), F32(np.sin(F32(angle)))
    ox = F32(((W - 1) - (c * (W - 1) - s * (H - 1))) / F32(2.0))
    oy = F32(((H - 1) - (s * (W - 1) + c * (H - 1))) / F32(2.0))
    ys, xs = np.meshgrid(np.arange(H, dtype=F32), np.arange(W, dtype=F32), indexing='ij')
    sx = _round_half_away((c * xs - s * ys + ox).astype(F32))
    sy = _round_half_away((s * xs + c * ys + oy).astype(F32))
    ok = (sx >= 0) & (sx < W) & (sy >= 0) & (sy < H)
    out = np.zeros_like(img)
    out[ok] = img[sy[ok], sx[ok]]
    return out


def resize_nearest(img, out_h, out_w):
    img = np.asarray(img, F32)
    in_h, in_w = img.shape
    sy, sx = F32(in_h) / F32(out_h), F32(in_w) / F32(out_w)
    yy = np.minimum(np.floor(np.arange(out_h, dtype=F32) * sy).astype(np.int64), in_h - 1)
    xx = np.minimum(np.floor(np.arange(out_w, dtype=F32) * sx).astype(np.int64), in_w - 1)
    return img[yy][:, xx]


def crop_or_pad(img, th, tw):
    """tf.image.resize_image_with_crop_or_pad."""
    h, w = img.shape
    dh, dw = th - h, tw - w
    ch, cw = max(-dh // 2, 0), max(-dw // 2, 0)
    ph, pw = max(dh // 2, 0), max(dw // 2, 0)
    img = img[ch:ch + min(th, h), cw:cw + min(tw, w)]
    out = np.zeros((th, tw), F32)
    out[ph:ph + img.shape[0], pw:pw + img.shape[1]] = img
    return out


def data_aug_one(dm, pose, cfg, com, angle, ratio_h, ratio_w):
    """preprocess.py:235-262 with the three random draws injected: angle ~ U(-pi, pi),
    (ratio_h, ratio_w) = clip(N(1, 0.2), 0.9, 1.1).  Returns (aug_dm, aug_pose)."""
    dm = np.asarray(dm, F32)
    H, W = dm.shape
    rot = rotate_nearest(dm, angle)
    uv_com = pro(com, cfg).reshape(3)
    uvd = pro(pose, cfg) - uv_com[None]
    c, s = F32(np.cos(F32(angle))), F32(np.sin(F32(angle)))
    rot_mat = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], F32)
    rp = (uvd @ rot_mat).astype(F32)
    th, tw = int(F32(H) * F32(ratio_h)), int(F32(W) * F32(ratio_w))
    out = crop_or_pad(resize_nearest(rot, th, tw), H, W)
    rp = rp * np.array([ratio_w, ratio_h, 1.0], F32)[None] + uv_com[None]
    return out, bpro(rp, cfg).reshape(-1).astype(F32)


def data_aug(dms, poses, cfgs, coms, params):
    """Batch form; params[b] = (angle, ratio_h, ratio_w)."""
    outs = [data_aug_one(dms[b], poses[b], cfgs[b], coms[b], *params[b]) for b in range(len(dms))]
    return np.stack([o[0] for o in outs]), np.stack([o[1] for o in outs])


def draw_aug_params(rng, B):
    """The reference's random draws (preprocess.py:238,252) with a numpy generator."""
    angle = rng.uniform(-np.pi, np.pi, B).astype(F32)
    ratios = np.clip(rng.normal(1.0, 0.2, (B, 2)), 0.9, 1.1).astype(F32)
    return np.concatenate([angle[:, None], ratios], 1).astype(F32)
