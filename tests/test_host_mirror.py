"""Host-side mirror of the reference's CLI / model interface / result-file format (CPU tests + one GPU run)."""
import math
import os

import numpy as np
import pytest

from tests.common import GOLDEN


def test_flags_match_reference_names_and_defaults():
    from densereg_amd import flags
    F = flags.parse([])
    # model/hourglass_um_crop_tiny.py:29-60
    assert (F.num_gpus, F.batch_size, F.debug_level, F.sub_batch, F.pid) == (1, 40, 1, 5, 0)
    assert (F.is_train, F.net_module, F.is_aug, F.dataset, F.epoch) == (True, 'um_v1', True, 'nyu', 80)
    assert (F.num_stack, F.num_fea, F.kernel_size) == (2, 128, 3)
    F = flags.parse(['--dataset', 'icvl', '--fea_num', '64', '--num_stack', '1', '--is_train', 'False', '--batch_size', '8'])
    assert F.num_fea == 64 and F.num_stack == 1 and F.is_train is False and F.dataset == 'icvl'      # README spelling accepted
    F = flags.parse(['--num_fea', '256', '--is_train', 'True'])
    assert F.num_fea == 256 and F.is_train is True
    # this build's additions keep the reference's behaviour by default
    F = flags.parse([])
    assert (F.precision, F.data_dir, F.restore_step, F.save_every) == ('f32', '', 0, 0)
    F = flags.parse(['--precision', 'bf16', '--data_dir', '/data/nyu'])
    assert F.precision == 'bf16' and F.data_dir == '/data/nyu'
    with pytest.raises(SystemExit):
        flags.parse(['--precision', 'fp8'])
    flags.parse([])


def test_checkpoint_a_test_run_restores(tmp_path):
    from densereg_amd.model.hourglass_um_crop_tiny import checkpoint_for_test
    d = str(tmp_path)
    assert checkpoint_for_test(d, 0) is None                                   # nothing published here: random weights
    open(os.path.join(d, 'model.ckpt--1.index'), 'wb').close()
    assert checkpoint_for_test(d, 0) == os.path.join(d, 'model.ckpt--1')      # the reference's run_test(…, -1)
    assert checkpoint_for_test(d, 1500) == os.path.join(d, 'model.ckpt-1500')  # explicit step
    assert checkpoint_for_test(d, -1) == os.path.join(d, 'model.ckpt--1')


def test_result_line_format_matches_shipped_prediction_files():
    """exp/result/icvl.txt / nyu.txt pin the output FORMAT: name, tab, %.4f fields, '\\' separators."""
    from densereg_amd.model.hourglass_um_crop_tiny import result_line
    for fname, J in (('icvl_result_head.txt', 16), ('nyu_result_head.txt', 14)):
        for line in open(os.path.join(GOLDEN, fname)):
            fields = line.rstrip('\n').split('\t')
            assert len(fields) == 1 + 3 * J
            vals = np.array([float(v) for v in fields[1:]])
            name_fwd = fields[0].replace('\\', '/')                  # what the dataset reader would hand over
            assert result_line(name_fwd, vals) == line              # byte-identical re-serialisation
    assert result_line('a/b.png', [1, 2.00004, -3.5]) == 'a\\b.png\t1.0000\t2.0000\t-3.5000\n'


def test_lr_schedule_and_model_naming():
    from densereg_amd import flags
    from densereg_amd.parallel import decay_steps, learning_rate
    # hourglass_um_crop_tiny.py:109,174 + train_single_gpu.py:45-49: nyu: 73730/(40*5)*10 = 3686.5 steps
    assert abs(decay_steps('nyu', 40, 5) - 3686.5) < 1e-9
    assert abs(decay_steps('msra', 40, 5) - 68085 / 200.0 * 20) < 1e-9
    assert learning_rate(0, 'nyu', 40, 5) == 1e-3 and learning_rate(3686, 'nyu', 40, 5) == 1e-3
    assert math.isclose(learning_rate(3687, 'nyu', 40, 5), 1e-4) and math.isclose(learning_rate(7373, 'nyu', 40, 5), 1e-5)
    flags.parse([])


def test_global_batch_is_split_over_ranks_like_the_reference_towers():
    """train_multi_gpu.py:58-62: --batch_size is the GLOBAL batch, split evenly over num_gpus; epochs / max_steps / the
    learning-rate staircase are those of the single-GPU run (BASELINE config 4: MSRA batch 320 on 8 GPUs = 40 per rank)."""
    from densereg_amd import flags
    from densereg_amd.model import hourglass_um_crop_tiny as M
    from densereg_amd.parallel import DataParallelTrainer, check_world, decay_steps, learning_rate, per_rank_batch
    assert per_rank_batch(320, 8) == 40 and per_rank_batch(40, 1) == 40
    with pytest.raises(ValueError, match='divisible'):
        per_rank_batch(40, 3)
    check_world(8, 8)
    with pytest.raises(ValueError, match='--num_gpus 1 but the launcher started 8'):
        check_world(1, 8)
    # the trainer's staircase counts optimizer steps of the global batch: 8 ranks x 40 == 1 rank x 320
    assert abs(decay_steps('msra', 320, 5) - 68085 / 1600.0 * 20) < 1e-9

    class _Dist:
        @staticmethod
        def get_world_size():
            return 8

        @staticmethod
        def get_rank():
            return 3
    calls = []

    class _Eng:
        def flat_view(self, which): return None
        def zero_grad(self): pass
        def apply_adam(self, lr, div, step, clip): calls.append((lr, div, step))
    tr = DataParallelTrainer(_Eng(), dataset='msra', sub_batch=5, dist=_Dist(), all_reduce=lambda g: None)
    assert (tr.world, tr.rank) == (8, 3)
    tr.global_step = int(decay_steps('msra', 320, 5)) + 1           # just past the first decay of the GLOBAL schedule
    tr.optimizer_step(40)
    assert math.isclose(calls[0][0], 1e-4) and calls[0][1] == 40.0 and calls[0][2] == tr.global_step
    assert math.isclose(learning_rate(tr.global_step - 1, 'msra', 320, 5), 1e-4)
    assert learning_rate(tr.global_step - 1, 'msra', 40, 5) == 1e-3        # what a per-rank-batch schedule would have said
    # model class: input / map sides follow --in_hw, max_steps from the global batch
    flags.parse(['--dataset', 'msra', '--batch_size', '320', '--num_gpus', '8', '--in_hw', '256'])
    try:
        class _DS:
            name, subset, jnt_num, approximate_num = 'msra', 'training', 21, 68085
        made = {}
        orig = M.um_v1.get_engine
        M.um_v1.get_engine = lambda J, hw, mb, dev, tr_: made.setdefault('args', (J, hw, mb, dev, tr_))
        model = M.JointDetectionModel(_DS(), None, epoch=80, world=8)
        assert made['args'][:3] == (21, 256, 40) and model.rank_batch == 40
        assert model._input_height == 256 and model._output_height == 64
        assert model.max_steps == int(80 * 68085 / 1600.0)
        assert abs(model.decay_steps - 68085 / 1600.0 * 20) < 1e-9
    finally:
        M.um_v1.get_engine = orig
        flags.parse([])


def test_evaluation_metrics_and_curve(tmp_path, capsys):
    """data/evaluation.py:9-18 and the curve file of :63-103, against a HAND-COMPUTED expected file: 17 thresholds
    5t+0.5, strict '<' for the curve, '<=' for the four printed shares, '%f %f\\n' with the share x 100."""
    from densereg_amd.data.evaluation import Evaluation
    a, b = np.zeros(6), np.array([3, 4, 0, 0, 0, 12.0])
    assert Evaluation.maxJntError(a, b) == 12.0 and Evaluation.meanJntError(a, b) == 8.5     # evaluation.py:9-18
    scores = [50.0, 1.0, 10.6, 5.0, 80.5, 10.5]                     # unsorted on purpose (:64 sorts)
    th, frac = Evaluation.plotError(scores, str(tmp_path / 'e.txt'))
    below = [0, 2, 2, 4, 4, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5, 5]     # scores strictly below 0.5, 5.5, 10.5, ... 80.5
    expected = ''.join('%f %f\n' % (5.0 * t + 0.5, n / 6.0 * 100.0) for t, n in enumerate(below))
    assert open(tmp_path / 'e.txt').read() == expected
    assert expected.splitlines()[0] == '0.500000 0.000000' and expected.splitlines()[2] == '10.500000 33.333333'
    assert expected.splitlines()[-1] == '80.500000 83.333333' and len(expected.splitlines()) == 17
    assert th == [5.0 * t + 0.5 for t in range(17)] and frac[3] == 4 / 6.0
    out = capsys.readouterr().out.splitlines()                      # 10.5 counts for '<=' but not for '<'
    assert out == ['10mm percentage: 0.500000', '20mm percentage: 0.666667', '30mm percentage: 0.666667', '40mm percentage: 0.666667']
    th2, frac2 = Evaluation.averageMaxJntError(scores, log=None)    # :21-61 returns the same curve
    assert th2 == th and frac2 == frac


def test_synthetic_dataset_constants():
    from densereg_amd.data.synthetic import DATASETS, make_crops
    assert [DATASETS[k]['jnt_num'] for k in ('icvl', 'nyu', 'msra')] == [16, 14, 21]          # icvl.py:17, nyu.py:40-45, msra.py:17
    assert DATASETS['icvl']['exact_num'] == 1596 and DATASETS['nyu']['exact_num'] == 8252
    dm, pose, cfg, com, names = make_crops(3, 'nyu', seed=1)
    dm2 = make_crops(3, 'nyu', seed=1)[0]
    np.testing.assert_array_equal(dm, dm2)
    assert dm.shape == (3, 128, 128, 1) and pose.shape == (3, 42) and cfg.shape == (3, 6) and com.shape == (3, 3)
    assert 0.25 < (dm > 0).mean() < 0.6 and np.all(cfg[:, 4:] == 128)


def test_learnable_synthetic_hands_are_a_function_of_the_image():
    """``make_hand_crops`` (the crops the engine trains itself on, tests/test_trained_parity.py): seeded and reproducible, every joint
    projects onto the hand it belongs to (the palm centre and points along the five fingers, 6 mm behind the visible surface), the
    centre of mass follows data/preprocess.py:131-142, and -- unlike ``make_crops`` -- two crops with the same geometry give the
    same joints (the joints are not random foreground pixels)."""
    from densereg_amd.data.synthetic import DATASETS, center_of_mass, make_hand_crops
    for dataset in ('icvl', 'nyu', 'msra'):
        J = DATASETS[dataset]['jnt_num']
        dm, pose, cfg, com, names = make_hand_crops(6, dataset, seed=3)
        dm2, pose2, cfg2, com2, _ = make_hand_crops(6, dataset, seed=3)
        np.testing.assert_array_equal(dm, dm2)
        np.testing.assert_array_equal(pose, pose2)
        assert dm.shape == (6, 128, 128, 1) and pose.shape == (6, 3 * J) and len(names) == 6
        other = make_hand_crops(6, dataset, seed=4)[1]
        assert np.abs(other - pose).max() > 1.0                        # another seed, another hand
        for b in range(6):
            img, j = dm[b, :, :, 0], pose[b].reshape(J, 3)
            fg = img > 0
            assert 0.08 < fg.mean() < 0.35
            np.testing.assert_allclose(com[b], center_of_mass(img, cfg[b]), rtol=1e-6)
            u = j[:, 0] * cfg[b, 0] / j[:, 2] + cfg[b, 2]
            v = j[:, 1] * cfg[b, 1] / j[:, 2] + cfg[b, 3]
            iu, iv = np.clip(np.round(u).astype(int), 0, 127), np.clip(np.round(v).astype(int), 0, 127)
            assert fg[iv, iu].all(), (dataset, b)                      # every joint lies on the silhouette ...
            np.testing.assert_allclose(j[:, 2], img[iv, iu] + 6.0, atol=1e-3)      # ... 6 mm behind the surface the camera sees
            # fingers fan out from the palm: the joints of one finger (j = 1 + f + 5 s) move away from the palm centre with s
            d = np.hypot(u - u[0], v - v[0])
            for f in range(5):
                chain = d[1 + f::5]
                assert (np.diff(chain) > 0).all(), (dataset, b, f)


def test_accumulation_window_as_one_pass_policy_and_bookkeeping():
    """parallel.window_groups picks the pass size; DataParallelTrainer.window_step does the bookkeeping of `sub_batch` micro-steps
    and the optimizer step around ONE forward / loss / backward of the engine in groups mode."""
    from densereg_amd.parallel import DataParallelTrainer, window_groups
    assert window_groups(40, 5, 128) == 5                    # BASELINE configs 2-4: 40 crops x 5 micro-steps on 32x32 maps
    assert window_groups(40, 5, 256) == 1                    # config 5: 64x64 maps, the window would be 819 200 pixels per layer
    assert window_groups(40, 1, 128) == 1 and window_groups(40, 9, 128) == 1 and window_groups(4, 2, 128) == 1
    assert window_groups(40, 5, 128, override=0) == 1 and window_groups(40, 5, 256, override=5) == 5
    with pytest.raises(ValueError):
        window_groups(4, 2, 128, override=2)                 # 4 crops per micro-batch: the 2x2 layers cannot be cut into tiles
    with pytest.raises(ValueError):
        window_groups(40, 5, 128, override=3)
    calls = []

    class _T:
        def __init__(self, n): self.shape = (n,)
        def reshape(self, *s): calls.append(('reshape', s)); return self

    class _Eng:
        pipeline = 1
        def flat_view(self, which): return None
        def zero_grad(self): calls.append('zero')
        def set_groups(self, g): calls.append(('groups', g))
        def forward_train(self, dm, mode, mask, seed): calls.append(('fwd', dm.shape[0], seed))
        def loss(self, dm, pose, cfg, com): calls.append('loss'); return _T(20)
        def backward(self, B): calls.append(('bwd', B))
        def apply_adam(self, lr, div, step, clip): calls.append(('adam', div, step))
    tr = DataParallelTrainer(_Eng(), dataset='nyu', sub_batch=5)
    calls.clear()
    tr.window_step(_T(200), None, None, None, seed=10)
    assert calls == [('groups', 5), ('fwd', 200, 10), 'loss', ('bwd', 200), ('groups', 1), ('adam', 5.0, 1), 'zero', ('reshape', (5, 4))]
    assert tr.micro == 5 and tr.global_step == 1
    with pytest.raises(ValueError):
        tr.window_step(_T(201), None, None, None)


@pytest.mark.gpu
def test_cli_test_and_train_drivers_on_gpu(gpu, tmp_path, monkeypatch):
    from densereg_amd import flags
    from densereg_amd.model import hourglass_um_crop_tiny as M
    monkeypatch.chdir(tmp_path)
    flags.parse(['--dataset', 'icvl', '--num_stack', '1', '--fea_num', '64', '--is_train', 'False', '--batch_size', '8',
                 '--num_frames', '20'])
    ds, vs = M.SyntheticDataset('icvl', 'training'), M.SyntheticDataset('icvl', 'testing')
    from densereg_amd.network import um_v1
    eng = um_v1.get_engine(16, 128, 8, 0, False)
    eng.load_params(M._random_params(eng))
    model, (max_err, mean_err), out = M.run_test(ds, vs)
    lines = open(out).read().splitlines()
    assert len(lines) == 20 and all(len(l.split('\t')) == 49 for l in lines) and '\\' in lines[0] and '/' not in lines[0]
    assert model.name == 'icvl_training_s1_f64_daug_um_v1'            # hourglass_um_crop_tiny.py:115-117,534-535
    # two optimizer steps of the training driver
    flags.parse(['--dataset', 'nyu', '--num_stack', '1', '--fea_num', '64', '--is_train', 'True', '--batch_size', '4',
                 '--sub_batch', '2', '--max_steps', '2'])
    ds = M.SyntheticDataset('nyu', 'training')
    eng = um_v1.get_engine(14, 128, 4, 0, True)
    eng.load_params(M._random_params(eng))
    before = eng.read_params()
    model, trainer = M.run_train(ds, None)
    after = eng.read_params()
    assert trainer.global_step == 2 and model.window_groups == 1
    assert any(np.abs(after[k] - before[k]).max() > 0 for k in before if k.endswith('weights'))
    # ... and with micro-batches the engine can run as groups: each accumulation window is one pass (parallel.window_groups)
    flags.parse(['--dataset', 'nyu', '--num_stack', '1', '--fea_num', '64', '--is_train', 'True', '--batch_size', '8',
                 '--sub_batch', '2', '--max_steps', '2'])
    eng = um_v1.get_engine(14, 128, 16, 0, True)
    eng.load_params(M._random_params(eng))
    before = eng.read_params()
    model, trainer = M.run_train(ds, None)
    after = eng.read_params()
    assert model.window_groups == 2 and model.engine is eng and trainer.global_step == 2 and trainer.micro == 4
    assert any(np.abs(after[k] - before[k]).max() > 0 for k in before if k.endswith('weights'))
    assert all(np.isfinite(v).all() for v in after.values())
    flags.parse([])


@pytest.mark.gpu
def test_engine_checkpoint_round_trip_with_adam(gpu, tmp_path):
    """Engine.save_checkpoint / load_checkpoint (TF V2 files with the reference's names): after two optimizer
    steps, a fresh engine restored from the files continues bit-identically (variables, BatchReNorm slots, Adam)."""
    import torch
    from densereg_amd import checkpoint as ck
    from densereg_amd.data.synthetic import make_crops
    from densereg_amd.engine import Engine
    from densereg_amd.parallel import DataParallelTrainer
    B = 4
    dm, poses, cfgs, coms, _ = make_crops(B, 'nyu', seed=3)

    def fresh():
        e = Engine(num_stack=1, num_fea=32, num_jnt=14, max_batch=B, training=True)
        rng = np.random.default_rng(7)
        params = {}
        for name, shape, _ in e.param_infos():
            leaf = name.rsplit('/', 1)[1]
            if leaf == 'weights':
                fan = int(np.prod(shape[:3]))
                params[name] = (rng.standard_normal(shape) * np.sqrt(2.0 / fan)).astype(np.float32)
            elif leaf in ('gamma', 'moving_variance', 'r_max'):
                params[name] = np.ones(shape, np.float32)
            else:
                params[name] = np.zeros(shape, np.float32)
        e.load_params(params)
        return e

    t = [torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda() for a in (dm, poses, cfgs, coms)]

    def steps(e, trainer, n):
        normed = e.norm_dm(t[0], t[3])
        for i in range(n):
            trainer.micro_step(normed, t[1], t[2], t[3], seed=i, dropout_mode=0)
        torch.cuda.synchronize()

    a = fresh()
    ta = DataParallelTrainer(a, 'nyu', sub_batch=1)
    steps(a, ta, 2)
    prefix = str(tmp_path / 'model.ckpt-2')
    names = a.save_checkpoint(prefix, global_step=2)
    # everything Saver(tf.global_variables()).restore looks up by name: Adam's non-slot variables hold beta^(t+1) after
    # t updates, global_step is float32 (train_single_gpu.py:42)
    assert {'beta1_power', 'beta2_power', 'global_step'} <= set(names)
    from densereg_amd import checkpoint as _ck
    sc = _ck.read_checkpoint(prefix, names=['beta1_power', 'beta2_power', 'global_step'])
    assert sc['global_step'].dtype == np.float32 and float(sc['global_step']) == 2.0
    assert abs(float(sc['beta1_power']) - 0.5 ** 3) < 1e-7 and abs(float(sc['beta2_power']) - 0.999 ** 3) < 1e-7
    assert 'hg_imgproc/Conv/weights/Adam_1' in names and 'global_step' in names
    assert int(ck.read_checkpoint(prefix, names=['global_step'])['global_step']) == 2
    b = fresh()
    rep = b.load_checkpoint(prefix)
    assert rep['missing'] == [] and rep['unexpected'] == []
    tb = DataParallelTrainer(b, 'nyu', sub_batch=1)
    tb.global_step = 2
    pa, pb = a.read_params(), b.read_params()
    for k in pa:
        np.testing.assert_array_equal(pa[k], pb[k], err_msg=k)
    steps(a, ta, 1)
    steps(b, tb, 1)
    pa, pb = a.read_params(), b.read_params()
    worst = max(float(np.abs(pa[k] - pb[k]).max() / (np.abs(pa[k]).max() + 1e-12)) for k in pa)
    # same state in, same step out -- up to the order of the few fp atomics left on the path (max-pool backward
    # scatter, stem moments, loss sums), which this network's backward amplifies: run-to-run gradient noise is up to
    # 1e-4 of a tensor's max (see test_lanes_match_single_stream); measured here 1e-6 .. 9e-5
    assert worst < 1e-3, worst
    a.close(); b.close()


def test_training_driver_keeps_one_window_queued_ahead_of_the_loss_check(monkeypatch, tmp_path):
    """``train()`` (train_single_gpu.py:138-158 re-ordered): window k+1 is QUEUED before the losses of window k are read (the only
    device -> host synchronisation), a checkpoint is written only after its own window's losses were checked, a NaN loss still stops
    the run (one window late), and the producer thread ends with the loop.  A stand-in trainer records the order of events."""
    import io
    import threading

    import torch
    from densereg_amd import flags
    from densereg_amd.model import hourglass_um_crop_tiny as M
    events = []

    class _Losses:
        """losses of one window 'on the device': reading them is an event"""
        def __init__(self, k, bad=False): self.k, self.bad = k, bad
        def reshape(self, *s): return self
        def sum(self, dim=1): return self
        def tolist(self):
            events.append(('read', self.k))
            return [float('nan') if self.bad else 1.0] * 2

    class _Trainer:
        def __init__(self, eng, dataset, sub_batch, dist=None):
            self.global_step, self.k, self.nan_at = 0, 0, getattr(eng, 'nan_at', -1)
        def window_step(self, dm, pose, cfg, com, seed=0):
            events.append(('queue', self.k, int(dm.shape[0]), seed))
            self.k += 1
            self.global_step += 1
            return _Losses(self.k - 1, bad=(self.k - 1 == self.nan_at))

    class _Eng:
        def norm_dm(self, dm, com): return dm
        def save_checkpoint(self, path, global_step): events.append(('save', global_step))

    class _DS:
        name = 'nyu'
        def batch(self, bs, index):
            return (np.full((bs, 4, 4, 1), index, np.float32), np.zeros((bs, 42), np.float32), np.zeros((bs, 6), np.float32),
                    np.zeros((bs, 3), np.float32), ['f'] * bs)

    class _Model:
        engine, _dataset, device = _Eng(), _DS(), torch.device('cpu')
        rank_batch, window_groups, decay_steps, lr_decay_factor, init_lr, max_steps = 8, 2, 1.0, 0.1, 1e-3, 4
        train_dir = str(tmp_path)
        def _t(self, a): return torch.from_numpy(np.ascontiguousarray(a, np.float32))

    monkeypatch.setattr(M, 'DataParallelTrainer', _Trainer)
    flags.parse(['--dataset', 'nyu', '--is_train', 'True', '--batch_size', '8', '--sub_batch', '2', '--max_steps', '4', '--is_aug', 'False',
                 '--save_every', '2'])
    try:
        log = io.StringIO()
        M.train(_Model(), log=log)
        # every window holds its two micro-batches (seeds 0, 2, 4, 6 name the first micro-step of each); window 1 is queued BEFORE the
        # losses of window 0 are read; the checkpoint of step 2 comes after the losses of window 1 (its own), not before
        assert events == [('queue', 0, 16, 0), ('queue', 1, 16, 2), ('read', 0), ('read', 1), ('save', 2),
                          ('queue', 2, 16, 4), ('queue', 3, 16, 6), ('read', 2), ('read', 3), ('save', 4)], events
        assert 'crops/s end to end' in log.getvalue()
        # a diverged window stops the run when its losses are read, i.e. after the next window was queued -- and nothing is saved on top of it
        events.clear()
        m = _Model()
        m.engine.nan_at = 1
        with pytest.raises(AssertionError, match='NaN'):
            M.train(m, log=None)
        assert ('queue', 1, 16, 2) in events and ('read', 1) in events and not any(e[0] == 'save' for e in events)
        assert not any(t.name == 'densereg-batches' and t.is_alive() for t in threading.enumerate())
    finally:
        flags.parse([])
