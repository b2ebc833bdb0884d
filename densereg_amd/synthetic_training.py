"""Train the engine on the learnable synthetic crops (``data.synthetic.make_hand_crops``) with the reference's recipe: windows of
``sub_batch`` x 40 crops, Adam(0.5, 0.999), lr 1e-3, +-0.2 gradient clip, dropout on (model/train_single_gpu.py:45-89,138-150).

Used by ``examples/train_synthetic.py`` and by ``tests/test_trained_parity.py``: the parity of the vote against the oracle is a
statement about PEAKED heat-maps (what a trained model produces, readme.md:24-25) -- random weights give flat maps whose arg-max is
a coin toss between far-apart pixels.  No dataset and no checkpoint exist in this image, so the engine trains its own."""
from __future__ import annotations

import numpy as np
import torch

from .data.synthetic import DATASETS, make_hand_crops
from .engine import Engine
from .parallel import DataParallelTrainer


def reference_init(eng: Engine, seed: int = 11):
    """The reference's initialisation: conv weights truncated-normal(0.01) (slim/ops.py:272), biases / beta / moving mean 0, gamma /
    moving variance / r_max 1 (ops.py:87-128)."""
    rng = np.random.default_rng(seed)
    params = {}
    for name, shape, _ in eng.param_infos():
        leaf = name.rsplit('/', 1)[1]
        if leaf == 'weights':
            params[name] = (np.clip(rng.standard_normal(shape), -2, 2) * 0.01).astype(np.float32)
        elif leaf in ('gamma', 'moving_variance', 'r_max'):
            params[name] = np.ones(shape, np.float32)
        else:
            params[name] = np.zeros(shape, np.float32)
    return params


def joint_error_mm(xyz: np.ndarray, gt: np.ndarray) -> np.ndarray:
    """per-frame, per-joint Euclidean error (data/evaluation.py:9-18 takes its mean)"""
    d = xyz.reshape(xyz.shape[0], -1, 3) - gt.reshape(gt.shape[0], -1, 3)
    return np.sqrt((d ** 2).sum(-1))


def evaluate(eng_params, S, F, dataset, crops, device=0, batch=40):
    """forward(eval) + vote of an inference engine holding ``eng_params`` on ``crops`` = (dm, pose, cfg, com); returns xyz (N, 3J)."""
    dm, _pose, cfg, com = crops
    J = DATASETS[dataset]['jnt_num']
    ieng = Engine(S, F, J, dm.shape[1], 3, batch, device, training=False)
    ieng.load_params(eng_params)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(ieng.device)
    out = []
    for i in range(0, dm.shape[0], batch):
        d_com = t(com[i:i + batch])
        xyz = ieng.infer(ieng.norm_dm(t(dm[i:i + batch]), d_com), t(cfg[i:i + batch]), d_com)
        out.append(xyz.cpu().numpy())
    ieng.close()
    return np.concatenate(out)


def train(S=2, F=128, dataset='icvl', steps=200, train_crops=2000, sub_batch=5, micro=40, seed=7, device=0, log=None, lr_scale=1.0):
    """``steps`` optimizer steps over a fixed set of ``train_crops`` synthetic hands (cycled, reshuffled every epoch).
    Returns (params, loss history [steps, sub_batch, 4])."""
    J = DATASETS[dataset]['jnt_num']
    W = sub_batch * micro
    if train_crops < W:
        raise ValueError('train_crops=%d is less than one accumulation window of sub_batch x micro = %d crops' % (train_crops, W))
    train_crops -= train_crops % W                  # whole windows only (a ragged tail would never be drawn)
    eng = Engine(S, F, J, 128, 3, W, device, training=True)
    eng.load_params(reference_init(eng, seed))
    trainer = DataParallelTrainer(eng, dataset=dataset, sub_batch=sub_batch)
    dm, pose, cfg, com, _ = make_hand_crops(train_crops, dataset, seed=seed)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    d_com, d_pose, d_cfg = t(com), t(pose), t(cfg)
    d_dm = torch.cat([eng.norm_dm(t(dm[i:i + W]), d_com[i:i + W]) for i in range(0, train_crops, W)])
    rng = np.random.default_rng(seed)
    hist = []
    perm = None
    per_epoch = train_crops // W
    for step in range(steps):
        if step % per_epoch == 0:
            perm = torch.from_numpy(rng.permutation(train_crops)).to(eng.device)
        idx = perm[(step % per_epoch) * W:(step % per_epoch + 1) * W]
        losses = trainer.window_step(d_dm[idx].contiguous(), d_pose[idx].contiguous(), d_cfg[idx].contiguous(), d_com[idx].contiguous(),
                                     seed=step, dropout_mode=2)
        hist.append(losses)
        if log is not None and (step % 20 == 0 or step == steps - 1):
            lo = losses.cpu().numpy()
            log('step %4d  hm %.4f  hm3 %.4f  um %.4f  reg %.4f' % (step, lo[:, 0].mean(), lo[:, 1].mean(), lo[:, 2].mean(), lo[:, 3].mean()))
    torch.cuda.synchronize(eng.device)
    hist = torch.stack(hist).cpu().numpy()
    params = eng.read_params()
    eng.close()
    return params, hist
