"""Host-side mirror of ``model/hourglass_um_crop_tiny.py`` (reference): CLI flags, ``JointDetectionModel``
and the train / test drivers, on top of the HIP engine.

    python -m densereg_amd.model.hourglass_um_crop_tiny --dataset icvl --num_stack 2 --num_fea 128 --is_train False
    torchrun --nproc-per-node 8 -m densereg_amd.model.hourglass_um_crop_tiny --dataset msra --is_train True --num_gpus 8

What maps to what
  JointDetectionModel.test (:442-462)        -> ``test``      norm_dm + forward(eval) + vote, last stack only
  JointDetectionModel.loss (:323-371)        -> ``loss``      targets + 3 l2 losses per stack + L2 regulariser
  JointDetectionModel._xyz_estimation (:743) -> ``_xyz_estimation`` (the vote kernel)
  train_single_gpu.train (:37-177)           -> ``train``     accumulate sub_batch micro-steps, clip, Adam, LR staircase
  test_model.test (:14-94)                   -> ``test_model`` result file ``name\\t%.4f...`` with '/' -> '\\'
The reference's datasets and checkpoints are not distributable with this repo: the drivers run on the
seeded synthetic crop generator (``densereg_amd.data.synthetic``) and random-initialised weights unless
``load_params`` is given real ones.  Data augmentation (``--is_aug``, hourglass_um_crop_tiny.py:332-333) is applied
on the device by ``dr_data_aug`` inside the training loop and recorded in the model name.

Multi-GPU (``--num_gpus N`` under ``torchrun --nproc-per-node N``): ``--batch_size`` is the GLOBAL minibatch and is split
evenly over the ranks, as the reference splits it over its towers (train_multi_gpu.py:58-62); epochs, ``max_steps`` and the
learning-rate staircase are therefore the single-GPU ones.  ``--in_hw 256`` / ``512`` selects the larger crops the network
accepts (um_v1.py:99-104) and the reference model class does not (…tiny.py:82-87).
"""
from __future__ import annotations

import os
import sys
import time
from datetime import datetime

import numpy as np
import torch

from .. import flags
from ..data import preprocess
from ..data.evaluation import Evaluation
from ..data.synthetic import DATASETS, make_crops
from ..network import um_v1
from ..parallel import DataParallelTrainer, check_world, decay_steps, per_rank_batch, window_groups


class SyntheticDataset:
    """Stand-in for data.icvl/nyu/msra: camera, joint count, sizes; batches come from make_crops.  A FINITE set like the record files:
    batch ``index`` holds the seeded crops of ``index % (synthetic_crops // batch_size)``, synthesised once (2.8 ms of numpy per crop
    on the host: 350 crops/s against the 3200 the training step consumes) and kept -- later epochs cost a dictionary lookup."""

    def __init__(self, name: str, subset: str, rank: int = 0, hw: int = 128):
        ds = DATASETS[name]
        self.name, self.subset, self.rank, self.hw = name, subset, rank, hw
        self.jnt_num = ds['jnt_num']
        self.cfg = (ds['fx'], ds['fy'], ds['cx'], ds['cy'], ds['w'], ds['h'])
        self.approximate_num = ds['approximate_num']
        self.exact_num = ds['exact_num']
        self._kept = {}

    def batch(self, batch_size: int, index: int):
        period = max(1, int(getattr(flags.FLAGS, 'synthetic_crops', 4000)) // batch_size)
        key = (batch_size, index % period, flags.FLAGS.seed)
        hit = self._kept.get(key)
        if hit is None:
            hit = make_crops(batch_size, self.name, seed=flags.FLAGS.seed + 7919 * key[1], rank=self.rank, hw=self.hw)
            self._kept[key] = hit
        return hit


class JointDetectionModel(object):
    _init_lr = 0.001                 # :69
    _lr_decay_factor = 0.1           # :74
    _adam_beta1 = 0.5                # :76
    _input_height = _input_width = 128      # :82-83 (class defaults; --in_hw overrides them per instance)
    _output_height = _output_width = 32     # :86-87
    _base_dir = './exp/train_cache/'        # :92

    def __init__(self, dataset, detect_net, epoch, net_desc='dummy', val_dataset=None, device: int = 0, world: int = 1):
        F = flags.FLAGS
        self._dataset, self._val_dataset = dataset, val_dataset
        self._input_height = self._input_width = int(getattr(F, 'in_hw', 128))
        self._output_height = self._output_width = self._input_height // 4        # um_v1.py:109
        self._world = int(world)
        self._rank_batch = per_rank_batch(F.batch_size, self._world)                # train_multi_gpu.py:58-62
        self._jnt_num = int(dataset.jnt_num)
        self._net, self._net_desc = detect_net, net_desc
        self._num_batches_per_epoch = dataset.approximate_num / float(F.batch_size * F.sub_batch)     # :109
        self._max_steps = int(epoch * self._num_batches_per_epoch)                                    # :112
        self._model_desc = '%s_%s_s%d_f%d' % (dataset.name, dataset.subset, F.num_stack, F.num_fea)    # :115
        if F.is_aug:
            self._model_desc += '_daug'
        self.device = torch.device('cuda', device)
        # training: the micro-steps of an accumulation window as one pass of launches where that pays (parallel.window_groups)
        self._groups = window_groups(self._rank_batch, F.sub_batch, self._input_height, getattr(F, 'groups', -1)) if F.is_train else 1
        self.engine = um_v1.get_engine(self._jnt_num, self._input_height, self._rank_batch * self._groups, device, bool(F.is_train))

    # ---- hyper-parameters (:159-182) -----------------------------------------------------------
    @property
    def init_lr(self):
        return self._init_lr

    @property
    def lr_decay_factor(self):
        return self._lr_decay_factor

    @property
    def decay_steps(self):
        return decay_steps(self._dataset.name, flags.FLAGS.batch_size, flags.FLAGS.sub_batch)

    @property
    def max_steps(self):
        return self._max_steps

    @property
    def rank_batch(self):
        """crops per micro-step on THIS rank: the global --batch_size split over the ranks"""
        return self._rank_batch

    @property
    def window_groups(self):
        """micro-steps per pass of launches (1, or --sub_batch: the whole accumulation window)"""
        return self._groups

    @property
    def name(self):
        return '%s_%s' % (self._model_desc, self._net_desc)              # :534-535

    @property
    def train_dir(self):
        return os.path.join(self._base_dir, self.name)                    # :538-539

    # ---- the path ---------------------------------------------------------------------------------
    def _t(self, a):
        if isinstance(a, torch.Tensor):                                     # record datasets hand over device tensors
            return a.to(self.device, torch.float32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(self.device)

    def inference(self, normed_dms, cfgs, coms, is_training=True):
        return self._net(normed_dms, cfgs, coms, self._jnt_num, is_training, engine=self.engine)

    def _xyz_estimation(self, hms, ums, hm3s, dms, cfgs, coms):
        """(:743-785) -- takes the UNIT offset maps and the depth at input resolution; ``_resume_om`` (:276)
        and ``unnorm_xyz_pose`` (:462) are inside the kernel.  Returns xyz in mm, (B, 3J)."""
        return self.engine.vote(hms, hm3s, ums, dms, cfgs, coms)

    def test(self, dms, poses, cfgs, coms):
        """(:442-462) depth (mm) -> xyz (mm)."""
        normed = self.engine.norm_dm(dms, coms)
        return self.engine.infer(normed, cfgs, coms)

    def loss(self, dms, poses, cfgs, coms, seed=0):
        """(:323-371) one training micro-step's forward + loss; returns the 4 terms hm, hm3, um, reg."""
        normed = self.engine.norm_dm(dms, coms)
        self.engine.forward_train(normed, seed=seed)
        return self.engine.loss(normed, poses, cfgs, coms), normed


def result_line(name: str, xyz) -> str:
    """test_model.py:73-76."""
    s = '%s\t%s\n' % (name, '\t'.join(format(float(pt), '.4f') for pt in xyz))
    return s.replace('/', '\\')


def train(model: JointDetectionModel, dist=None, log=sys.stdout):
    """train_single_gpu.train (:37-177) with the data-parallel reduction of SURVEY 8(e)."""
    F = flags.FLAGS
    trainer = DataParallelTrainer(model.engine, dataset=model._dataset.name, sub_batch=F.sub_batch, dist=dist)
    max_steps = F.max_steps or model.max_steps
    if log:
        print('[train] learning rate decays per %d steps with rate=%f' % (model.decay_steps, model.lr_decay_factor), file=log)
        print('[train] initial learning_rate = %f' % model.init_lr, file=log)
    micro = 0
    start_step = 0
    if F.restore_step > 0:                                                  # train_single_gpu.py:120-123
        rep = model.engine.load_checkpoint(os.path.join(model.train_dir, 'model.ckpt-%d' % F.restore_step))
        start_step = F.restore_step
        trainer.global_step = start_step                                     # lr schedule and Adam's bias correction
        if log:
            print('[train] restored step %d (%d unexpected names in the checkpoint)' % (start_step, len(rep['unexpected'])), file=log)
    aug_rng = np.random.default_rng(F.seed + (dist.get_rank() if dist is not None else 0))
    # The reference's step (train_single_gpu.py:138-158) is "sub_batch x (fetch a batch, sess.run, assert not NaN), sess.run(train_op)":
    # the host waits for every micro-step's loss before it prepares the next batch.  Here the device must never wait for the host:
    #   * a producer thread prepares the host side of window k+1 (crop synthesis / record decode: _dataset.batch) while the device runs
    #     window k -- the queue runs one window ahead (the queue-runner role of data/dataset_base.py);
    #   * the losses of window k are READ (the only device -> host synchronisation of the loop) after window k+1 has been queued; the
    #     NaN assert of :147 therefore fires one window late -- one optimizer step has been applied on top of the diverged one, nothing
    #     is saved in between (the checkpoint of a step is written after its losses were checked).
    import queue
    import threading
    todo = queue.Queue(maxsize=2)
    stop = threading.Event()

    def produce():
        m = start_step * F.sub_batch
        try:
            if model.device.type == 'cuda':
                torch.cuda.set_device(model.device)                         # (record datasets decode on the device: this thread's current device)
            for _step in range(start_step, max_steps):
                if stop.is_set():
                    return
                todo.put([model._dataset.batch(model.rank_batch, m + k) for k in range(F.sub_batch)])
                m += F.sub_batch
        except BaseException as e:                                          # (the consumer re-raises it)
            todo.put(e)

    producer = threading.Thread(target=produce, name='densereg-batches', daemon=True)
    producer.start()
    micro = start_step * F.sub_batch
    pending = None                                                          # (step, start time, losses on the device) of the window in flight
    t_first = t_steady = None
    # windows left out of the steady-state rate: the first launches, and the first epoch of a synthetic set (its crops are being synthesised)
    n_warm = 2 + (-(-int(getattr(F, 'synthetic_crops', 0)) // (model.rank_batch * F.sub_batch)) if isinstance(model._dataset, SyntheticDataset) else 0)

    def settle(entry):
        """read a finished window's losses: the asserts and the log line of the reference's loop"""
        st, t0, losses, n_crops = entry
        ave_loss = 0.0
        for loss_value in losses.reshape(-1, 4).sum(dim=1).tolist():        # (blocks until that window's launches are done)
            assert not np.isnan(loss_value), 'Model diverged with loss = NaN'          # :147
            ave_loss += loss_value
        ave_loss /= F.sub_batch
        duration = time.time() - t0
        if log and st % 5 == 0:
            print('[model/train] %s: step %d/%d, loss = %.3f, %.3f sec/batch, %.3f sec/sample'
                  % (datetime.now(), st, max_steps, ave_loss, duration, duration / (F.batch_size * F.sub_batch)), file=log)

    try:
        for step in range(start_step, max_steps):
            host = todo.get()
            if isinstance(host, BaseException):
                raise host
            start = time.time()
            if t_first is None:
                t_first = start
            if step == start_step + n_warm:
                if model.device.type == 'cuda':
                    torch.cuda.synchronize(model.device)
                t_steady = time.time()
            window, step_losses = [], []
            for dm, poses, cfgs, coms, _n in host:
                d_dm, d_pose, d_cfg, d_com = model._t(dm), model._t(poses), model._t(cfgs), model._t(coms)
                if F.is_aug:                                                # hourglass_um_crop_tiny.py:332-333
                    d_dm, d_pose = preprocess.data_aug(d_dm, d_pose, d_cfg, d_com, generator=aug_rng)
                normed = model.engine.norm_dm(d_dm, d_com)
                if model.window_groups > 1:                                 # the window runs as one pass once it is complete
                    window.append((normed, d_pose, d_cfg, d_com))
                else:
                    step_losses.append(trainer.micro_step(normed, d_pose, d_cfg, d_com, seed=micro).reshape(4))
                micro += 1
            if window:
                parts = [torch.cat([w[k] for w in window]) for k in range(4)]
                losses = trainer.window_step(*parts, seed=micro - F.sub_batch)         # [sub_batch, 4]: one row per micro-step
            else:
                losses = torch.stack(step_losses)
            if pending is not None:
                settle(pending)                                             # window k-1, while window k runs
            pending = (step, start, losses, F.batch_size * F.sub_batch)
            if F.save_every > 0 and ((step + 1) % F.save_every == 0 or step + 1 == max_steps):   # :170-172
                settle(pending)                                             # a checkpoint is written after ITS losses were checked
                pending = None
                if dist is None or dist.get_rank() == 0:
                    os.makedirs(model.train_dir, exist_ok=True)
                    model.engine.save_checkpoint(os.path.join(model.train_dir, 'model.ckpt-%d' % (step + 1)), global_step=step + 1)
        if pending is not None:
            settle(pending)
    finally:
        stop.set()
        while producer.is_alive():                                          # unblock a producer waiting on a full queue
            try:
                todo.get_nowait()
            except queue.Empty:
                pass
            producer.join(0.05)
    if log and t_first is not None and max_steps > start_step:
        if model.device.type == 'cuda':
            torch.cuda.synchronize(model.device)
        el = time.time() - t_first
        per_step = F.batch_size * F.sub_batch // (dist.get_world_size() if dist is not None else 1)
        n = (max_steps - start_step) * per_step
        msg = '[train] %d optimizer steps, %d crops on this rank in %.3f s: %.1f crops/s end to end (host batches + copies + augmentation + step)' % (
            max_steps - start_step, n, el, n / el)
        if t_steady is not None and max_steps - start_step > n_warm:
            ns = (max_steps - start_step - n_warm) * per_step
            msg += '; after the first %d windows: %.1f crops/s' % (n_warm, ns / (time.time() - t_steady))
        print(msg, file=log)
    return trainer


def checkpoint_for_test(train_dir: str, restore_step: int):
    """Which checkpoint a test run restores.  The reference always restores step -1 (``run_test(dataset, val_dataset, -1)``,
    ``…tiny.py:908`` -> ``model.ckpt--1``, the name of the published models, ``exp/scripts/fetch_*_model.sh``); here
    ``--restore_step N`` (N != 0) names a step explicitly and must exist, and without it ``model.ckpt--1`` is restored when
    present -- otherwise the run keeps its random weights (no checkpoint can exist in this environment)."""
    if restore_step != 0:
        return os.path.join(train_dir, 'model.ckpt-%d' % restore_step)
    default = os.path.join(train_dir, 'model.ckpt--1')
    return default if os.path.exists(default + '.index') else None


def test_model(model: JointDetectionModel, out_path: str, log=sys.stdout):
    """test_model.test (:14-94): run the test set, write one result line per frame, return the errors."""
    F = flags.FLAGS
    total = F.num_frames or model._val_dataset.exact_num
    ckpt = checkpoint_for_test(model.train_dir, F.restore_step)             # test_model.py:31-35
    if ckpt is not None:
        model.engine.load_checkpoint(ckpt, strict=True)
        if log:
            print('[test_model]model has been restored from %s' % ckpt, file=log)
    max_err, mean_err, n, step = [], [], 0, 0
    with open(out_path, 'w') as f:
        while n < total:
            try:
                dm, poses, cfgs, coms, names = model._val_dataset.batch(model.rank_batch, step)
            except StopIteration:                                           # a record dataset shorter than exact_num
                break
            xyz = model.test(model._t(dm), model._t(poses), model._t(cfgs), model._t(coms)).cpu().numpy()
            for xyz_val, gt_val, name in zip(xyz, poses, names):
                max_err.append(Evaluation.maxJntError(xyz_val, gt_val))
                mean_err.append(Evaluation.meanJntError(xyz_val, gt_val))
                f.write(result_line(name, xyz_val))
                n += 1
                if n >= total:
                    break
            step += 1
    Evaluation.plotError(max_err, out_path.replace('.txt', '') + '_error.txt')
    if log:
        print('finish test: %d frames, mean joint error %.3f mm (random weights unless parameters were loaded)'
              % (n, float(np.mean(mean_err))), file=log)
    return max_err, mean_err


def run_train(dataset, val_dataset, dist=None, device=0):
    net_module = __import__('densereg_amd.network.' + flags.FLAGS.net_module, fromlist=['detect_net'])      # :863-867
    world = dist.get_world_size() if dist is not None else 1
    check_world(flags.FLAGS.num_gpus, world)
    model = JointDetectionModel(dataset, net_module.detect_net, epoch=flags.FLAGS.epoch, net_desc=net_module.TOWER_NAME,
                                val_dataset=val_dataset, device=device, world=world)
    return model, train(model, dist)


def run_test(train_dataset, test_dataset, out_path=None, device=0):
    net_module = __import__('densereg_amd.network.' + flags.FLAGS.net_module, fromlist=['detect_net'])      # :874-878
    model = JointDetectionModel(train_dataset, net_module.detect_net, epoch=flags.FLAGS.epoch, net_desc=net_module.TOWER_NAME,
                                val_dataset=test_dataset, device=device)
    os.makedirs(model.train_dir, exist_ok=True)
    out_path = out_path or os.path.join(model.train_dir, '%s-%s-result.txt' % (test_dataset.subset, datetime.now().strftime('%Y-%m-%d_%H:%M:%S')))
    return model, test_model(model, out_path), out_path


def _random_params(engine, seed=7):
    rng = np.random.default_rng(seed)
    params = {}
    for name, shape, _ in engine.param_infos():
        leaf = name.rsplit('/', 1)[1]
        if leaf == 'weights':
            params[name] = np.clip(rng.standard_normal(shape), -2, 2).astype(np.float32) * 0.01      # ops.py:272 trunc-normal 0.01
        elif leaf in ('gamma', 'moving_variance', 'r_max'):
            params[name] = np.ones(shape, np.float32)
        else:
            params[name] = np.zeros(shape, np.float32)
    return params


def main(argv=None):
    F = flags.parse(argv)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0')) % max(torch.cuda.device_count(), 1)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world)
    torch.cuda.set_device(local)
    if F.data_dir:                                                          # the reference's TFRecord shards (data/datasets.py)
        from ..data.datasets import get_dataset
        dataset = get_dataset(F.dataset, 'training', F.data_dir, F.pid)
        val_dataset = get_dataset(F.dataset, 'testing', F.data_dir, F.pid)
        dataset.rank, dataset.world, dataset.seed = rank, world, F.seed
    else:
        dataset = SyntheticDataset(F.dataset, 'training', rank, F.in_hw)
        val_dataset = SyntheticDataset(F.dataset, 'testing', rank, F.in_hw)
    if F.is_train:
        check_world(F.num_gpus, world)
    # the engine the model class will ask for (same cache key AND capacity: a training window that runs as one pass needs
    # rank_batch x sub_batch rows, JointDetectionModel.__init__) gets its random-init weights here
    rb = per_rank_batch(F.batch_size, world if F.is_train else 1)
    cap = rb * (window_groups(rb, F.sub_batch, F.in_hw, getattr(F, 'groups', -1)) if F.is_train else 1)
    eng = um_v1.get_engine(dataset.jnt_num, F.in_hw, cap, local, bool(F.is_train))
    eng.load_params(_random_params(eng))
    if F.is_train:
        run_train(dataset, val_dataset, dist, local)
    else:
        _, (max_err, mean_err), out = run_test(dataset, val_dataset, device=local)
        print('results written to %s' % out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
