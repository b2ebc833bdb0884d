"""Host mirror of the reference's input front-end (``data/preprocess.py``) over the C ABI.

Same function names and argument meaning as the reference, batched over device tensors; every function is one
launch of a HIP kernel in ``libdensereg_hip.so`` (``densereg_amd/csrc/frontend.h``) -- there is no torch or CPU
implementation behind them.

* ``crop_from_xyz_pose``   data/preprocess.py:10-79
* ``crop_from_bbx``        data/preprocess.py:81-129
* ``center_of_mass``       data/preprocess.py:131-142
* ``data_aug``             data/preprocess.py:234-268
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from .. import _lib


class FrontEndError(RuntimeError):
    pass


def _p(t: torch.Tensor):
    assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32, 'front-end tensors: contiguous fp32 device tensors'
    return t.data_ptr()


def _stream(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _check(rc, what):
    if rc != 0:
        raise FrontEndError('%s failed with code %d' % (what, rc))


def crop_and_com_from_pose(dms, poses, cfgs, out_w, out_h, pad=20.0, dataset='nyu'):
    """crop_from_xyz_pose + center_of_mass in one launch -> (crops (B,h,w), poses, new_cfgs (B,6), coms (B,3))."""
    if out_w != out_h:
        raise ValueError('square crops only (the reference always uses 128x128)')
    B, H, W = dms.shape
    J = poses.shape[1] // 3
    crops = torch.empty(B, out_h, out_w, dtype=torch.float32, device=dms.device)
    new_cfgs = torch.empty(B, 6, dtype=torch.float32, device=dms.device)
    coms = torch.empty(B, 3, dtype=torch.float32, device=dms.device)
    with torch.cuda.device(dms.device):
        rc = _lib.load().dr_crop_from_pose(B, _p(dms), H, W, _p(poses), J, _p(cfgs), int(dataset == 'icvl'), float(pad), out_h,
                                           _p(crops), _p(new_cfgs), _p(coms), _stream(dms))
    _check(rc, 'dr_crop_from_pose')
    return crops, poses, new_cfgs, coms


def crop_from_xyz_pose(dms, poses, cfgs, out_w, out_h, pad=20.0, dataset='nyu'):
    """Reference signature (batched): -> [cropped_dms, poses, new_cfgs]."""
    crops, poses, new_cfgs, _ = crop_and_com_from_pose(dms, poses, cfgs, out_w, out_h, pad, dataset)
    return [crops, poses, new_cfgs]


def crop_and_com_from_bbx(dms, poses, bbxs, cfgs, out_w, out_h):
    if out_w != out_h:
        raise ValueError('square crops only (the reference always uses 128x128)')
    B, H, W = dms.shape
    crops = torch.empty(B, out_h, out_w, dtype=torch.float32, device=dms.device)
    new_cfgs = torch.empty(B, 6, dtype=torch.float32, device=dms.device)
    coms = torch.empty(B, 3, dtype=torch.float32, device=dms.device)
    with torch.cuda.device(dms.device):
        rc = _lib.load().dr_crop_from_bbx(B, _p(dms), H, W, _p(bbxs), _p(cfgs), out_h, _p(crops), _p(new_cfgs), _p(coms),
                                          _stream(dms))
    _check(rc, 'dr_crop_from_bbx')
    return crops, poses, new_cfgs, coms


def crop_from_bbx(dms, poses, bbxs, cfgs, out_w, out_h):
    crops, poses, new_cfgs, _ = crop_and_com_from_bbx(dms, poses, bbxs, cfgs, out_w, out_h)
    return [crops, poses, new_cfgs]


def center_of_mass(dms, cfgs):
    """Centre of mass of already cropped square maps: the crop kernel with the whole map as its box (scale 1:
    the resize is the identity) -- only its com output is used."""
    B, H, W = dms.shape
    if H != W:
        raise ValueError('center_of_mass expects the square crops the network consumes')
    bbx = torch.tensor([0.0, 0.0, float(H), float(W), 3.0e38], dtype=torch.float32, device=dms.device).repeat(B, 1).contiguous()
    return crop_and_com_from_bbx(dms, None, bbx, cfgs, W, H)[3]


def draw_aug_params(B, generator: np.random.Generator):
    """The reference's draws: angle ~ U(-pi, pi), edge ratios = clip(N(1, 0.2), 0.9, 1.1) (preprocess.py:238,252)."""
    angle = generator.uniform(-math.pi, math.pi, B).astype(np.float32)
    ratios = np.clip(generator.normal(1.0, 0.2, (B, 2)), 0.9, 1.1).astype(np.float32)
    return np.concatenate([angle[:, None], ratios], 1).astype(np.float32)


def data_aug(dms, poses, cfgs, coms, draws=None, generator=None):
    """-> (aug_dms, aug_poses).  ``draws`` (B,3) = angle, ratio_h, ratio_w; drawn on the host if omitted."""
    squeeze = dms.dim() == 4
    d3 = dms.reshape(dms.shape[0], dms.shape[1], dms.shape[2]) if squeeze else dms
    B, H, W = d3.shape
    J = poses.shape[1] // 3
    if draws is None:
        draws = torch.from_numpy(draw_aug_params(B, generator or np.random.default_rng())).to(dms.device)
    out = torch.empty_like(d3)
    out_pose = torch.empty_like(poses)
    with torch.cuda.device(dms.device):
        rc = _lib.load().dr_data_aug(B, _p(d3), H, W, _p(poses), J, _p(cfgs), _p(coms), _p(draws), _p(out), _p(out_pose),
                                     _stream(dms))
    _check(rc, 'dr_data_aug')
    return (out.reshape(dms.shape) if squeeze else out), out_pose
