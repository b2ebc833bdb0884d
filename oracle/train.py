"""Loss, gradients (torch autograd) and the optimizer step (TEST INFRASTRUCTURE).

Parity unpinned (see oracle/__init__.py).

* ``loss_and_grads``  model/hourglass_um_crop_tiny.py:323-371 -- per stack
  l2_loss(hm-gt)+l2_loss(hm3-gt)+l2_loss(um-gt) with l2_loss = sum(x^2)/2 (summed, not
  averaged) plus the L2 regulariser; gradients by autograd through the restated graph
  with r/d of BatchReNorm stop-gradiented.
* ``adam_step``       model/train_single_gpu.py:83-89 + hourglass_um_crop_tiny.py:436-439 --
  g = clip(acc/sub_batch, -0.2, 0.2); Adam(beta1=0.5, beta2=0.999, eps=1e-8) with
  [TF1.3-semantics] lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t*m/(sqrt(v)+eps).
* ``learning_rate``   model/train_single_gpu.py:45-49 (staircase exponential decay).
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

from .graph import NetConfig, trainable_names
from .net import detect_net, reg_loss, to_torch_params
from .pose import make_targets

ADAM_BETA1 = 0.5      # hourglass_um_crop_tiny.py:76
ADAM_BETA2 = 0.999
ADAM_EPS = 1e-8
GRAD_CLIP = 0.2       # train_single_gpu.py:86


def loss_and_grads(cfg: NetConfig, params: Dict[str, np.ndarray], dm_norm: np.ndarray, poses: np.ndarray,
                   cfgs: np.ndarray, coms: np.ndarray, dropout_masks=None, dtype=torch.float32, conv_operands='f32', switches=None):
    """One micro-step. Returns (losses dict, grads dict name->np, bn_updates, end_points np).  ``switches``: oracle/net.py, TorchOps."""
    gt_hm, gt_hm3, gt_um = make_targets(dm_norm, poses, cfgs, coms, cfg.out_hw)
    tp = to_torch_params(params, dtype, requires_grad=True)
    dm = torch.from_numpy(dm_norm).to(dtype)
    masks = None if dropout_masks is None else [torch.from_numpy(m) for m in dropout_masks]
    ep, ops = detect_net(cfg, tp, dm, True, masks, conv_operands=conv_operands, switches=switches)
    tg = lambda a: torch.from_numpy(a).to(dtype)
    l2 = lambda t: (t * t).sum() * 0.5
    hm_loss = sum(l2(e - tg(gt_hm)) for e in ep['hm_outs'])
    hm3_loss = sum(l2(e - tg(gt_hm3)) for e in ep['hm3_outs'])
    um_loss = sum(l2(e - tg(gt_um)) for e in ep['um_outs'])
    reg = reg_loss(cfg, tp)
    total = reg + hm_loss + um_loss + hm3_loss
    total.backward()
    grads = {n: tp[n].grad.detach().numpy().astype(np.float64 if dtype == torch.float64 else np.float32)
             for n in trainable_names(cfg)}
    losses = {'hm': float(hm_loss), 'hm3': float(hm3_loss), 'um': float(um_loss), 'reg': float(reg),
              'total': float(total)}
    outs = {k: [t.detach().contiguous().numpy() for t in v] for k, v in ep.items()}
    return losses, grads, ops.bn_updates, outs


def learning_rate(step: int, init_lr: float, decay_steps: float, factor: float = 0.1) -> float:
    return init_lr * factor ** math.floor(step / decay_steps)


def adam_step(params, m, v, acc_grads, lr, t, div):
    """In-place Adam on float32 dicts.  ``div`` = sub_batch*world (tf.divide, :86); t = 1-based step."""
    lr_t = lr * math.sqrt(1.0 - ADAM_BETA2 ** t) / (1.0 - ADAM_BETA1 ** t)
    for n, g in acc_grads.items():
        g = np.clip(g.astype(np.float32) / np.float32(div), -GRAD_CLIP, GRAD_CLIP).astype(np.float32)
        m[n] = (np.float32(ADAM_BETA1) * m[n] + np.float32(1 - ADAM_BETA1) * g).astype(np.float32)
        v[n] = (np.float32(ADAM_BETA2) * v[n] + np.float32(1 - ADAM_BETA2) * g * g).astype(np.float32)
        params[n] = (params[n] - np.float32(lr_t) * m[n] / (np.sqrt(v[n]) + np.float32(ADAM_EPS))).astype(np.float32)
