"""Multi-process data-parallel step on CPU: world_size 2 over gloo, each rank drives its own engine
(the host-fiber emulation build of the kernels) through ``DataParallelTrainer``; the flat gradient is
all-reduced with ``torch.distributed`` exactly as bench.py does over RCCL.

Checks the semantics of SURVEY.md section 8(e): every rank ends the optimizer step with identical
parameters, equal to a single-process reference that accumulates both ranks' micro-batches and divides
by ``sub_batch * world``.
"""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.common import ROOT

SUB = 1
CFG = (1, 8, 2)       # num_stack, num_fea, num_jnt  (tiny heads are fixed-width: 128/256/512)
B = 1


class _EmuEngine:
    """The subset of densereg_amd.engine.Engine the trainer uses, over the emulation library."""

    def __init__(self):
        from tests.common import EmuBackend
        from oracle.graph import NetConfig
        self.be = EmuBackend()
        self.cfg = NetConfig(*CFG)
        self.h = self.be.handle(self.cfg, B, training=True)
        addr, n = self.h.flat('grad')
        import ctypes as C
        self._grad = torch.from_numpy(np.ctypeslib.as_array((C.c_float * n).from_address(addr)))

    def flat_view(self, which):
        assert which == 'grad'
        return self._grad

    def load_params(self, params):
        self.h.load_params(params)
        self.h.call('dr_finalize_params', None)

    def forward_train(self, dm, mode, keep_mask, seed):
        import ctypes as C
        self.h.call('dr_forward_train', dm.shape[0], dm.ctypes.data, int(mode), None, C.c_uint64(seed), None)

    def loss(self, dm, pose, cfg, com):
        out = np.zeros(4, np.float32)
        self.h.call('dr_loss', dm.shape[0], dm.ctypes.data, pose.ctypes.data, cfg.ctypes.data, com.ctypes.data,
                    out.ctypes.data, None)
        return out

    def backward(self, Bn):
        self.h.call('dr_backward', Bn, None)

    def zero_grad(self):
        self.h.call('dr_zero_grad', None)

    def apply_adam(self, lr, div, step, clip):
        import ctypes as C
        self.h.call('dr_apply_adam', C.c_float(lr), C.c_float(div), C.c_float(clip), C.c_int64(step), None)


def _data(rank):
    from densereg_amd.data.synthetic import make_crops
    from oracle import pose
    dm, poses, cfgs, coms, _ = make_crops(B, 'icvl', seed=300, rank=rank)
    poses = np.ascontiguousarray(poses[:, :3 * CFG[2]])
    return pose.norm_dm(dm, coms), poses, cfgs, coms


def _params():
    from oracle import net
    from oracle.graph import NetConfig
    return net.init_params(NetConfig(*CFG), 11)


def _worker(rank, world, init_file, out_dir):
    sys.path.insert(0, ROOT)
    dist.init_process_group('gloo', init_method='file://' + init_file, rank=rank, world_size=world)
    from densereg_amd.parallel import DataParallelTrainer
    eng = _EmuEngine()
    eng.load_params(_params())
    tr = DataParallelTrainer(eng, dataset='nyu', sub_batch=SUB, dist=dist)
    ndm, poses, cfgs, coms = _data(rank)
    for i in range(SUB):                    # SUB micro-steps = 1 optimizer step
        tr.micro_step(ndm, poses, cfgs, coms, seed=i, dropout_mode=0)
    assert tr.global_step == 1
    got = eng.h.read_params()
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), **{k.replace('/', '|'): v for k, v in got.items()})
    dist.destroy_process_group()


def _dropout_worker(rank, world, init_file, out_dir):
    """Both ranks get the SAME crops and parameters and run with DR_DROPOUT_RNG: whatever differs between them is the mask."""
    sys.path.insert(0, ROOT)
    dist.init_process_group('gloo', init_method='file://' + init_file, rank=rank, world_size=world)
    from densereg_amd.parallel import DataParallelTrainer
    from oracle.graph import conv_specs
    eng = _EmuEngine()
    eng.load_params(_params())
    tr = DataParallelTrainer(eng, dataset='nyu', sub_batch=2, dist=dist)
    ndm, poses, cfgs, coms = _data(0)
    name = [c.name for c in conv_specs(eng.cfg) if c.cout == 512 and not c.bn][0]          # um_full 1: bias + ReLU + dropout
    acts = []
    for i in range(2):                      # 2 micro-steps = 1 optimizer step
        tr.micro_step(ndm, poses, cfgs, coms, seed=i, dropout_mode=2)
        if i == 0:
            acts.append(eng.be.read_activation(eng.h, name, (B, 32, 32, 512)))
    assert tr.global_step == 1
    got = eng.h.read_params()
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), act=acts[0], **{k.replace('/', '|'): v for k, v in got.items()})
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_gloo_step_matches_single_process_reference():
    world = 2
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, 'rdzv')
        mp.spawn(_worker, args=(world, init_file, d), nprocs=world, join=True)
        r0 = dict(np.load(os.path.join(d, 'rank0.npz')))
        r1 = dict(np.load(os.path.join(d, 'rank1.npz')))
    from oracle.graph import NetConfig, trainable_names
    names = [n.replace('/', '|') for n in trainable_names(NetConfig(*CFG))]
    for k in names:
        np.testing.assert_array_equal(r0[k], r1[k], err_msg=k)        # replicas stay in lock-step
    # single-process reference: same engine code, both ranks' micro-batches accumulated, div = sub_batch*world
    eng = _EmuEngine()
    params = _params()
    eng.load_params(params)
    eng.zero_grad()
    # rank-local BatchReNorm statistics: each rank's forward sees only its own crops, and its state after
    # the first micro-step feeds its second one -- reproduce per rank on fresh engines, sum the gradients.
    total = None
    for rank in range(world):
        e = _EmuEngine()
        e.load_params(params)
        e.zero_grad()
        ndm, poses, cfgs, coms = _data(rank)
        for i in range(SUB):
            e.forward_train(ndm, 0, None, i)
            e.loss(ndm, poses, cfgs, coms)
            e.backward(B)
        g = e.flat_view('grad').clone()
        total = g if total is None else total + g
    eng.flat_view('grad').copy_(total)
    from densereg_amd.parallel import GRAD_CLIP, learning_rate
    eng.apply_adam(learning_rate(0, 'nyu', B * world, SUB), float(SUB * world), 1, GRAD_CLIP)      # the GLOBAL batch, as the trainer
    ref = eng.h.read_params()
    for k in names:
        np.testing.assert_allclose(r0[k], ref[k.replace('|', '/')], rtol=1e-6, atol=1e-7, err_msg=k)
    changed = sum(float(np.abs(r0[k] - params[k.replace('|', '/')]).max()) > 0 for k in names)
    assert changed > len(names) * 0.9


@pytest.mark.timeout(900)
def test_two_rank_dropout_masks_differ_and_replicas_stay_in_lock_step():
    """SURVEY 8(e) / train_multi_gpu.py:63-90: each tower draws its own dropout masks.  Two ranks fed IDENTICAL crops and
    parameters with DR_DROPOUT_RNG: the kept units of um_full1 must differ between the ranks (the seed carries the rank,
    parallel.rank_seed), about half are dropped on each, and after the all-reduced optimizer step both replicas hold
    identical parameters."""
    world = 2
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, 'rdzv')
        mp.spawn(_dropout_worker, args=(world, init_file, d), nprocs=world, join=True)
        r0 = dict(np.load(os.path.join(d, 'rank0.npz')))
        r1 = dict(np.load(os.path.join(d, 'rank1.npz')))
    a0, a1 = r0.pop('act'), r1.pop('act')
    live = (a0 != 0) | (a1 != 0)                       # units with a positive pre-activation that at least one rank kept
    assert live.sum() > 1000
    both = ((a0 != 0) & (a1 != 0)).sum() / live.sum()  # independent fair masks: P(both | at least one) = 1/3
    assert 0.25 < both < 0.42, both
    k = (a0 != 0) & (a1 != 0)
    np.testing.assert_array_equal(a0[k], a1[k])        # same inputs and weights: a unit both ranks kept has the same value
    from oracle.graph import NetConfig, trainable_names
    for n in (n.replace('/', '|') for n in trainable_names(NetConfig(*CFG))):
        np.testing.assert_array_equal(r0[n], r1[n], err_msg=n)


def test_rank_seed_is_injective_over_ranks_and_micro_steps():
    from densereg_amd.parallel import rank_seed
    seen = {rank_seed(m, r, 8) for m in range(50) for r in range(8)}
    assert len(seen) == 400
    assert rank_seed(7, 0, 1) == 7                      # one rank: the micro-step counter itself, as before


class _ToyEngine:
    """A stand-in with the trainer's engine interface and a 64-element flat gradient: backward adds a rank- and call-dependent
    ramp, apply_adam records what it was given.  (The kernels behind a window pass are covered by tests/test_groups.py; this is
    the N > 1 bookkeeping around it.)"""
    pipeline = 1

    def __init__(self, rank):
        self.rank, self.grad, self.calls, self.groups = rank, torch.zeros(64), [], 1

    def flat_view(self, which):
        return self.grad

    def zero_grad(self):
        self.grad.zero_()

    def set_groups(self, g):
        self.groups = g

    def forward_train(self, dm, mode, mask, seed):
        self.calls.append(('fwd', dm.shape[0], self.groups, seed))

    def loss(self, dm, pose, cfg, com):
        return torch.arange(4.0 * self.groups)

    def backward(self, B):
        self.grad += (self.rank + 1) * self.groups * torch.arange(64.0)          # what `groups` micro-steps accumulate

    def apply_adam(self, lr, div, step, clip):
        self.calls.append(('adam', self.grad.clone(), div, step))


def _window_worker(rank, world, init_file, out_dir):
    sys.path.insert(0, ROOT)
    dist.init_process_group('gloo', init_method='file://' + init_file, rank=rank, world_size=world)
    from densereg_amd.parallel import DataParallelTrainer
    eng = _ToyEngine(rank)
    tr = DataParallelTrainer(eng, dataset='nyu', sub_batch=5, dist=dist)
    dm = torch.zeros(5 * 8, 4, 4)
    losses = tr.window_step(dm, dm, dm, dm, seed=3, dropout_mode=2)
    assert tuple(losses.shape) == (5, 4) and tr.micro == 5 and tr.global_step == 1 and eng.groups == 1
    fwd = [c for c in eng.calls if c[0] == 'fwd']
    adam = [c for c in eng.calls if c[0] == 'adam']
    assert len(fwd) == 1 and fwd[0][1:3] == (40, 5) and len(adam) == 1
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), grad=adam[0][1].numpy(), div=adam[0][2], seed=fwd[0][3],
             after=eng.grad.numpy())
    dist.destroy_process_group()


def test_two_rank_window_step_all_reduces_once_per_window():
    """SURVEY 8(e) with the accumulation window as one pass (``DataParallelTrainer.window_step``): one all-reduce(sum) of the
    flat gradient per window, division by sub_batch x world inside the optimizer kernel, per-rank dropout seeds, accumulator
    cleared afterwards -- world 2 over gloo."""
    world = 2
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, 'rdzv')
        mp.spawn(_window_worker, args=(world, init_file, d), nprocs=world, join=True)
        r0 = dict(np.load(os.path.join(d, 'rank0.npz')))
        r1 = dict(np.load(os.path.join(d, 'rank1.npz')))
    want = (1 + 2) * 5 * np.arange(64.0)                 # both ranks' five micro-steps, summed
    np.testing.assert_array_equal(r0['grad'], want)
    np.testing.assert_array_equal(r1['grad'], want)
    assert float(r0['div']) == float(r1['div']) == 10.0
    assert int(r0['seed']) != int(r1['seed'])            # parallel.rank_seed
    assert not r0['after'].any() and not r1['after'].any()
