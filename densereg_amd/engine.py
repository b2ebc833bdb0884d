"""Torch-facing engine: device memory and streams come from PyTorch-ROCm, all compute from the HIP
library behind the C ABI (``include/densereg.h``).  There is no eager / CPU fallback here: every
method hands raw device pointers to ``libdensereg_hip.so``.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), 'engine tensors must be contiguous device tensors'
    return t.data_ptr()


def _f32(t: torch.Tensor) -> torch.Tensor:
    assert t.dtype == torch.float32
    return t


class Engine:
    """One engine per (process, device).  Mirrors the call sites of the reference:

    * ``forward_eval`` / ``forward_train``  <->  ``um_v1.detect_net(dm, cfgs, coms, J, is_training)``
    * ``vote`` / ``infer``                  <->  ``JointDetectionModel._xyz_estimation`` / ``.test``
    * ``loss`` / ``backward`` / ``apply_adam`` <-> ``.loss`` and the step of ``train_single_gpu.train``
    """

    def __init__(self, num_stack=2, num_fea=128, num_jnt=16, in_hw=128, kernel_size=3, max_batch=40,
                 device: int = 0, training: bool = False, pipeline: Optional[int] = None):
        """``pipeline`` (training engines): micro-steps in flight, ``dr_set_pipeline``.  2 lets the kernels of micro-step k+1 fill
        the launch boundaries and small-grid chains of micro-step k (measured on MI355X, B=40: S=2 F=128 fp32 2050 -> 2250 crops/s,
        bf16 3350 -> 4006, MSRA J=21 2021 -> 2215) and costs a second set of per-micro-step buffers.  Where every layer already
        runs many rounds of workgroups there is nothing to fill and two streams of big kernels only evict each other's L2 lines
        (S=4 F=256 on 256x256 crops, 163 840 pixels per layer: bf16 473 -> 394, fp32 222 -> 205), so the default (``None``;
        ``DR_PIPELINE=1|2`` overrides) is 2 up to 65 536 pixels per full-resolution layer (max_batch x map side squared) and 1
        above.  When the device cannot hold the second set the engine says so and runs at depth 1."""
        self.lib = _lib.load()
        self.device = torch.device('cuda', device)
        with torch.cuda.device(self.device):
            self.h = _lib.Handle(self.lib, num_stack, num_fea, num_jnt, in_hw, kernel_size, max_batch, device, training)
        self.pipeline = 1
        if training:
            auto = 2 if max_batch * (in_hw // 4) ** 2 <= 65536 else 1
            depth = int(os.environ.get('DR_PIPELINE', auto)) if pipeline is None else int(pipeline)
            if depth == 2:
                try:
                    with torch.cuda.device(self.device):
                        self.h.call('dr_set_pipeline', 2)
                    self.pipeline = 2
                except _lib.DenseRegError as e:
                    if e.code != -5:                                   # DR_E_NOMEM: a speed feature, not a correctness one
                        raise
                    sys.stderr.write('densereg_amd: no memory for a second micro-step slot, running one micro-step at a time (%s)\n' % e)
        self.num_stack, self.num_fea, self.num_jnt, self.in_hw = num_stack, num_fea, num_jnt, in_hw
        self.map_hw = in_hw // 4
        self.max_batch = max_batch
        self.training = training
        self.groups = 1

    # ---- plumbing ---------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        self.h.close()

    def set_precision(self, precision: str):
        """'f32' (default) or 'bf16': matrix-core arithmetic of every k != 7 convolution of the handle (dr_set_precision) --
        eval forward / infer, and on a training engine the train-mode forward, the input gradients and the weight
        gradients; BatchReNorm, loss, Adam and the master weights stay fp32.  Call before load_params (it un-finalizes
        the handle and the packed weights follow the precision)."""
        self.h.call('dr_set_precision', {'f32': 0, 'bf16': 1}[precision])

    def set_fusion(self, on: bool):
        """Eval mode: the part of every hourglass below 16x16 pixels as one launch (``dr_set_fusion``; on by default).  Off keeps
        every layer's output in HBM (``read_activation``)."""
        self.h.call('dr_set_fusion', 1 if on else 0)

    def load_params(self, params: Dict[str, np.ndarray]):
        self.h.load_params(params)
        self.h.call('dr_finalize_params', self._stream())

    def read_params(self) -> Dict[str, np.ndarray]:
        torch.cuda.synchronize(self.device)
        return self.h.read_params()

    def param_infos(self):
        return self.h.param_infos()

    def new(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    # ---- the path ---------------------------------------------------------------------------
    def norm_dm(self, dm_mm: torch.Tensor, com: torch.Tensor) -> torch.Tensor:
        out = torch.empty_like(dm_mm)
        self.h.call('dr_norm_dm', dm_mm.shape[0], _p(_f32(dm_mm)), _p(_f32(com)), _p(out), self._stream())
        return out

    def forward_eval(self, dm_norm: torch.Tensor, want_maps: bool = True):
        B, J, m = dm_norm.shape[0], self.num_jnt, self.map_hw
        hm = hm3 = um = None
        if want_maps:
            hm, hm3, um = self.new(B, m, m, J), self.new(B, m, m, J), self.new(B, m, m, 3 * J)
        self.h.call('dr_forward_eval', B, _p(_f32(dm_norm)), _p(hm), _p(hm3), _p(um), self._stream())
        return hm, hm3, um

    def read_maps(self, B: int, stack: int):
        J, m = self.num_jnt, self.map_hw
        hm, hm3, um = self.new(B, m, m, J), self.new(B, m, m, J), self.new(B, m, m, 3 * J)
        self.h.call('dr_read_maps', B, stack, _p(hm), _p(hm3), _p(um), self._stream())
        return hm, hm3, um

    def vote(self, hm, hm3, um, dm_norm, cfg, com) -> torch.Tensor:
        B = dm_norm.shape[0]
        xyz = self.new(B, 3 * self.num_jnt)
        self.h.call('dr_vote', B, _p(hm), _p(hm3), _p(um), _p(dm_norm), _p(cfg), _p(com), _p(xyz), self._stream())
        return xyz

    def infer(self, dm_norm, cfg, com, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        B = dm_norm.shape[0]
        xyz = out if out is not None else self.new(B, 3 * self.num_jnt)
        self.h.call('dr_infer', B, _p(dm_norm), _p(cfg), _p(com), _p(xyz), self._stream())
        return xyz

    def read_activation(self, scope: str, B: int, shape) -> np.ndarray:
        torch.cuda.synchronize(self.device)
        a = np.empty(shape, np.float32)
        self.h.call('dr_read_activation', scope.encode(), B, a.ctypes.data, a.size)
        return a

    # ---- training ---------------------------------------------------------------------------
    def forward_train(self, dm_norm, dropout_mode=_lib.DROPOUT_RNG, keep_mask: Optional[torch.Tensor] = None, seed: int = 0):
        self.h.call('dr_forward_train', dm_norm.shape[0], _p(_f32(dm_norm)), int(dropout_mode), _p(keep_mask),
                    C.c_uint64(seed), self._stream())

    def loss(self, dm_norm, pose_mm, cfg, com) -> torch.Tensor:
        """The four loss terms of the last ``forward_train``: a tensor of 4, or ``[groups, 4]`` (one row per micro-batch) after
        ``set_groups(G > 1)``."""
        out = self.new(4) if self.groups == 1 else self.new(self.groups, 4)
        self.h.call('dr_loss', dm_norm.shape[0], _p(dm_norm), _p(_f32(pose_mm)), _p(_f32(cfg)), _p(_f32(com)), _p(out),
                    self._stream())
        return out

    def backward(self, B: int):
        self.h.call('dr_backward', B, self._stream())

    def zero_grad(self):
        self.h.call('dr_zero_grad', self._stream())

    def set_pipeline(self, depth: int):
        """Micro-steps in flight (``dr_set_pipeline``): 1 or 2.  Going to 1 finishes what is in flight and folds the gradients."""
        self.h.call('dr_set_pipeline', int(depth))
        self.pipeline = int(depth)

    def set_groups(self, groups: int):
        """Micro-batch groups (``dr_set_groups``): the next ``forward_train`` / ``loss`` / ``backward`` take ``groups`` consecutive
        micro-batches of one accumulation window at once (per-micro-batch BatchReNorm statistics and state chain, summed
        gradient); 1 restores one micro-batch per call."""
        if int(groups) != self.groups:
            self.h.call('dr_set_groups', int(groups))
            self.groups = int(groups)

    def groups_supported(self, micro_batch: int, groups: int) -> bool:
        """Can ``groups`` micro-batches of ``micro_batch`` crops run as one pass on this engine?  (``dr_set_groups``: at most 8
        groups, whole 32-row tiles per micro-batch in the 2x2 layers = a multiple of 8 crops, and the buffers to hold them.)"""
        return (self.training and 1 <= groups <= 8 and micro_batch % 8 == 0 and micro_batch * groups <= self.max_batch)

    def sync_grads(self):
        """Order the current stream behind every micro-step in flight and make ``flat_view('grad')`` the sum of all slots'
        accumulated gradients (before an all-reduce, or before reading the view); nothing to do at pipeline depth 1."""
        self.h.call('dr_sync_grads', self._stream())

    def apply_adam(self, lr: float, div: float, step: int, clip: float = 0.2):
        self.h.call('dr_apply_adam', C.c_float(lr), C.c_float(div), C.c_float(clip), C.c_int64(step), self._stream())

    def flat_view(self, which: str) -> torch.Tensor:
        """Zero-copy torch view of the flat fp32 gradient / parameter buffer (for RCCL all-reduce)."""
        ptr, n = self.h.flat(which)
        return _as_tensor(ptr, n, self.device)

    def conv_flops_per_crop(self) -> float:
        return self.h.conv_flops_per_crop()

    # ---- checkpoints (tf.train.Saver files of the reference, train_single_gpu.py:108-123,172) -----
    def _trainable_offsets(self):
        off, out = 0, {}
        for name, shape, trainable in self.h.param_infos():
            if trainable:
                n = int(np.prod(shape))
                out[name] = (off, n)
                off += n
        return out

    def adam_views(self):
        m, v, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
        self.h.call('dr_flat_adam', C.byref(m), C.byref(v), C.byref(n))
        return _as_tensor(m.value, n.value, self.device), _as_tensor(v.value, n.value, self.device)

    def load_checkpoint(self, prefix: str, strict: bool = True) -> dict:
        """Restore from ``<prefix>.index`` / ``.data-*`` written by the reference (or by ``save_checkpoint``): model
        variables, BatchReNorm state incl. the zero-debias slots, and -- on a training engine -- Adam's moments.
        Returns the import report (``missing`` / ``unexpected`` names, ``scalars`` such as ``global_step``)."""
        from . import checkpoint
        torch.cuda.synchronize(self.device)
        rep = checkpoint.load_into(self.h, prefix, strict=strict)
        if self.training and (rep['adam_m'] or rep['adam_v']):
            m, v = self.adam_views()
            for src, dst in ((rep['adam_m'], m), (rep['adam_v'], v)):
                for name, (off, n) in self._trainable_offsets().items():
                    if name in src:
                        dst[off:off + n].copy_(torch.from_numpy(src[name].reshape(-1)))
        self.h.call('dr_finalize_params', self._stream())
        return rep

    def save_checkpoint(self, prefix: str, global_step: Optional[int] = None, beta_powers=None):
        """Write a checkpoint the reference's ``Saver.restore`` reads by name (variables, slots, Adam moments)."""
        from . import checkpoint
        torch.cuda.synchronize(self.device)
        extra = {}
        if self.training:
            m, v = (t.cpu().numpy() for t in self.adam_views())
            shapes = {n: s for n, s, _ in self.h.param_infos()}
            for name, (off, n) in self._trainable_offsets().items():
                extra[name + '/Adam'] = m[off:off + n].reshape(shapes[name])
                extra[name + '/Adam_1'] = v[off:off + n].reshape(shapes[name])
            # AdamOptimizer's non-slot variables: Saver(tf.global_variables()).restore needs them.  After t applied updates
            # they hold beta^(t+1) (TF initialises them to beta and multiplies once per apply_gradients).
            if beta_powers is None and global_step is not None:
                beta_powers = (0.5 ** (int(global_step) + 1), 0.999 ** (int(global_step) + 1))
            if beta_powers is not None:
                extra['beta1_power'] = np.array(beta_powers[0], np.float32)
                extra['beta2_power'] = np.array(beta_powers[1], np.float32)
        return checkpoint.export_from(self.h, prefix, global_step=global_step, extra=extra)


class _CudaArray:
    """``__cuda_array_interface__`` shim so torch can alias library-owned device memory."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {'shape': (n,), 'typestr': '<f4', 'data': (ptr, False), 'version': 2}


def _as_tensor(ptr: int, n: int, device) -> torch.Tensor:
    return torch.as_tensor(_CudaArray(ptr, n), device=device)
