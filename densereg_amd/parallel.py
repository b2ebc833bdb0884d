"""Data-parallel training step: one process per GPU, one RCCL all-reduce per optimizer step.

Semantics: the reference's working trainer accumulates gradients over
``sub_batch`` micro-batches, divides, clips to +-0.2 and applies Adam
(``model/train_single_gpu.py:69-89,144-150``); its multi-GPU sketch averages per-tower gradients
(``model/train_multi_gpu.py:16-39``) with per-tower BatchReNorm statistics (:63-64, 85-86).  Here every
rank owns a full replica, accumulates its local micro-batches into the engine's flat fp32 gradient
buffer, and once per optimizer step the flat buffer is all-reduced (sum) over RCCL/xGMI; the division
by ``sub_batch * world`` happens inside the fused clip+Adam kernel.  Crops are independent units, so
there is no other data-path collective; moving BatchReNorm statistics stay per-rank.
"""
from __future__ import annotations

import math
import os

from .data.synthetic import DATASETS

INIT_LR = 0.001            # hourglass_um_crop_tiny.py:69
LR_DECAY = 0.1             # :74
GRAD_CLIP = 0.2            # train_single_gpu.py:86


def per_rank_batch(global_batch: int, world: int) -> int:
    """``--batch_size`` is the GLOBAL minibatch, split evenly over the ranks like the reference splits it over its towers
    (train_multi_gpu.py:58-62: ``assert FLAGS.batch_size % FLAGS.num_gpus == 0`` + ``tf.split``)."""
    if world < 1 or global_batch % world != 0:
        raise ValueError('the batch_size should be divisible wrt num_gpus (batch_size=%d, num_gpus=%d)' % (global_batch, world))
    return global_batch // world


def check_world(num_gpus: int, world: int) -> None:
    """``--num_gpus`` must name the number of ranks the launcher started (one process per GPU)."""
    if num_gpus != world:
        raise ValueError('--num_gpus %d but the launcher started %d rank(s): launch with torchrun --nproc-per-node %d, '
                         'or pass --num_gpus %d' % (num_gpus, world, num_gpus, world))


def rank_seed(seed: int, rank: int, world: int) -> int:
    """Dropout seed of one rank's micro-step.  Every tower of the reference builds its OWN dropout ops
    (train_multi_gpu.py:63-90 calls the model once per tower; slim/ops.py:710-728 ``tf.nn.dropout`` draws an independent
    mask per op), so two towers never drop the same units of the same micro-step.  The engine's keep bit is a stateless
    function of (seed, layer, element): interleaving the ranks into the seed gives every (micro-step, rank) its own stream."""
    return int(seed) * int(world) + int(rank)


def decay_steps(dataset: str, batch_size: int, sub_batch: int) -> float:
    """hourglass_um_crop_tiny.py:109,174: (approximate_num / (batch*sub_batch)) * epochs_per_decay (a float).
    ``batch_size`` is the GLOBAL batch (all ranks together), as in the reference."""
    ds = DATASETS[dataset]
    return ds['approximate_num'] / float(batch_size * sub_batch) * ds['epochs_per_decay']


def learning_rate(step: int, dataset: str, batch_size: int, sub_batch: int) -> float:
    """tf.train.exponential_decay(staircase=True) of train_single_gpu.py:45-49."""
    return INIT_LR * LR_DECAY ** math.floor(step / decay_steps(dataset, batch_size, sub_batch))


def window_groups(micro_batch: int, sub_batch: int, in_hw: int, override=None) -> int:
    """How many micro-steps of an accumulation window run as ONE pass of launches (``dr_set_groups``): ``sub_batch`` (the whole
    window) or 1.  The pass needs 2..8 micro-batches of a multiple of 8 crops (include/densereg.h) and buffers for the whole
    window; by default it is used up to 262 144 pixels per full-resolution layer and window (5 x 40 crops on 32x32 maps: 204 800
    -- measured on MI355X: 2250 -> 2511 crops/s fp32, 3944 -> 4827 bf16; a 256x256-crop window would be 819 200 pixels and
    4x the activation memory for kernels that already run many rounds of workgroups).  ``override`` (``--groups`` /
    ``DR_GROUPS``): 0 or 1 = one micro-step per pass, ``sub_batch`` = the whole window or an error if it cannot be."""
    if override is not None and override < 0:
        override = None
    if override is None and os.environ.get('DR_GROUPS') not in (None, ''):
        override = int(os.environ['DR_GROUPS'])
    ok = 2 <= sub_batch <= 8 and micro_batch % 8 == 0
    if override is not None and override >= 0:
        if override <= 1:
            return 1
        if not ok or override != sub_batch:
            raise ValueError('groups=%d: needs groups == sub_batch (%d) in 2..8 and micro-batches of a multiple of 8 crops (%d)'
                             % (override, sub_batch, micro_batch))
        return sub_batch
    return sub_batch if ok and micro_batch * sub_batch * (in_hw // 4) ** 2 <= 262144 else 1


class DataParallelTrainer:
    def __init__(self, engine, dataset: str = 'nyu', sub_batch: int = 5, dist=None, all_reduce=None):
        """``micro_step`` takes this rank's share of the minibatch; the learning-rate staircase counts optimizer steps
        of the GLOBAL batch (``batch * world`` crops per micro-step), so a run on N ranks sees the same number of epochs
        and decays at the same point in the data as the single-GPU run of the reference."""
        self.eng = engine
        self.dataset = dataset
        self.sub_batch = int(sub_batch)
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self._all_reduce = all_reduce            # injectable (gloo tests)
        self.micro = 0
        self.global_step = 0                     # optimizer steps applied so far
        self.flat_grad = engine.flat_view('grad') if engine is not None else None
        if engine is not None:
            # an optimizer step after EVERY micro-step leaves nothing to overlap and the second slot's bookkeeping only costs
            # (measured: 1867 against 2054 crops/s): one micro-step at a time then
            if self.sub_batch < 2 and getattr(engine, 'pipeline', 1) == 2:
                engine.set_pipeline(1)
            engine.zero_grad()

    def reduce_gradients(self):
        if self.world > 1 or (self.dist is not None and os.environ.get('DR_FORCE_ALLREDUCE') == '1'):
            sync = getattr(self.eng, 'sync_grads', None)
            if sync is not None:
                sync()                       # micro-steps in flight finish and their slots' gradients become one sum (dr_sync_grads)
            if self._all_reduce is not None:
                self._all_reduce(self.flat_grad)
            else:
                self.dist.all_reduce(self.flat_grad, op=self.dist.ReduceOp.SUM)

    def micro_step(self, dm_norm, pose_mm, cfg, com, seed: int = 0, dropout_mode: int = 2, keep_mask=None):
        """One ``sess.run([accum_op, batchnorm_update_op, loss])`` (:146); returns the 4 loss terms (device).
        ``seed`` names the micro-step (the same on every rank); each rank drops its own units (``rank_seed``)."""
        eng = self.eng
        eng.forward_train(dm_norm, dropout_mode, keep_mask, rank_seed(seed, self.rank, self.world))
        losses = eng.loss(dm_norm, pose_mm, cfg, com)
        eng.backward(dm_norm.shape[0])
        self.micro += 1
        if self.micro % self.sub_batch == 0:
            self.optimizer_step(dm_norm.shape[0])
        return losses

    def window_step(self, dm_norm, pose_mm, cfg, com, seed: int = 0, dropout_mode: int = 2, keep_mask=None):
        """A whole accumulation window at once: the tensors hold ``sub_batch`` consecutive micro-batches, the engine runs them as
        micro-batch groups in ONE pass of launches (``Engine.set_groups``: per-micro-batch BatchReNorm statistics, the state
        chained in order, the gradient sum) and the optimizer step follows -- what ``sub_batch`` calls of ``micro_step`` do, with
        ``sub_batch`` times the rows per kernel launch.  Returns the loss terms ``[sub_batch, 4]`` (device).  The optimizer step has
        been applied when this returns: a caller that checks the losses for NaN (the reference asserts after every micro-step,
        train_single_gpu.py:147) sees them one update later than with ``micro_step``.  Windows the one-pass form cannot take
        (``Engine.groups_supported``) run as ``sub_batch`` micro-steps instead."""
        eng, G = self.eng, self.sub_batch
        B = dm_norm.shape[0]
        if B % G:
            raise ValueError('window_step: %d crops are not %d micro-batches' % (B, G))
        if self.micro % G:
            raise ValueError('window_step: %d micro-step(s) of an unfinished window are pending' % (self.micro % G))
        supported = getattr(eng, 'groups_supported', None)
        if supported is not None and not supported(B // G, G):
            # what the one-pass window cannot take (micro-batches that are not a multiple of 8 crops, a window larger than the
            # engine's max_batch, more than 8 micro-batches): the same window as G micro-steps, G passes of launches.  Same result
            # with dropout off or with an injected keep mask; with DR_DROPOUT_RNG the two paths draw DIFFERENT (equally valid) keep
            # bits -- one pass keys them by (rank_seed(seed), element of the 5x40-crop tensor), the loop by (seed*G+g, element of a
            # 40-crop tensor) -- so a run that switches paths is not bit-comparable across the switch
            Bg, out = B // G, []
            for g in range(G):
                sl = slice(g * Bg, (g + 1) * Bg)
                km = None if keep_mask is None else keep_mask[:, sl].contiguous()       # [dropout layer][crop][h][w][512]
                out.append(self.micro_step(dm_norm[sl], pose_mm[sl], cfg[sl], com[sl], seed=seed * G + g, dropout_mode=dropout_mode,
                                           keep_mask=km))
            import torch
            return torch.stack([o.reshape(4) for o in out])
        eng.set_groups(G)
        try:
            eng.forward_train(dm_norm, dropout_mode, keep_mask, rank_seed(seed, self.rank, self.world))
            losses = eng.loss(dm_norm, pose_mm, cfg, com)
            eng.backward(B)
        finally:
            eng.set_groups(1)
        self.micro += G
        self.optimizer_step(B // G)
        return losses.reshape(G, 4)

    def optimizer_step(self, batch_size: int):
        """``sess.run(train_op)`` (:150) then ``reset_op`` (:144)."""
        self.reduce_gradients()
        lr = learning_rate(self.global_step, self.dataset, batch_size * self.world, self.sub_batch)
        self.global_step += 1
        self.eng.apply_adam(lr, float(self.sub_batch * self.world), self.global_step, GRAD_CLIP)
        self.eng.zero_grad()
