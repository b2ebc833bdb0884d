"""The vote against the oracle on TRAINED weights -- BASELINE.json's "<= 0.1 mm mean-joint-error delta on identical inputs" over ALL
joints, no exclusions.

The reference's evidence is a trained model's predictions (readme.md:24-25, exp/result/*.txt); no checkpoint and no dataset exist in
this image, and on RANDOM weights the heat-maps are flat: the arg-max of the vote (hourglass_um_crop_tiny.py:598-785) is then a coin
toss between far-apart pixels and a last-bit difference in a map moves a joint by centimetres (tests/test_bench_shapes.py counts those
joints and ALSO shows the oracle doing the same to itself between fp32 and fp64).  So the engine trains its own weights first: the
reference's recipe (truncated-normal 0.01 init, windows of 5 x 40 crops, Adam(0.5), lr 1e-3, +-0.2 clip, dropout) on learnable
synthetic hands (densereg_amd/data/synthetic.py::make_hand_crops) until the maps are peaked -- which is itself the end-to-end test of
the training path: the loss has to fall and the held-out joint error has to drop by more than 10x.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

S, F, J, B, DATASET = 2, 128, 16, 40, 'icvl'
STEPS, TRAIN_CROPS = 400, 2000


@pytest.fixture(scope='module')
def trained(gpu):
    from densereg_amd.data.synthetic import make_hand_crops
    from densereg_amd.engine import Engine
    from densereg_amd.synthetic_training import evaluate, joint_error_mm, reference_init, train
    held = make_hand_crops(400, DATASET, seed=991)[:4]
    e0 = Engine(S, F, J, 128, 3, B, 0, training=False)
    init = reference_init(e0)
    e0.close()
    err0 = joint_error_mm(evaluate(init, S, F, DATASET, tuple(a[:80] for a in held)), held[1][:80]).mean()
    params, hist = train(S, F, DATASET, STEPS, TRAIN_CROPS)
    return dict(params=params, hist=hist, held=held, err0=float(err0))


def test_training_makes_the_loss_fall_and_the_heldout_error_drop(trained):
    """model/train_single_gpu.py:138-150 through the engine: the data terms of the loss fall by a large factor and the voted joints of
    HELD-OUT hands (never trained on) move from the untrained network's error to within centimetres of the truth."""
    from densereg_amd.synthetic_training import evaluate, joint_error_mm
    hist = trained['hist']                                   # [steps, 5 micro-batches, (hm, hm3, um, reg)]
    assert np.isfinite(hist).all()
    hm, hm3, um = (hist[:, :, k].mean(-1) for k in range(3))     # per optimizer step
    q = len(hm) // 4
    print('loss terms, mean over the four quarters of the run:  hm %s | hm3 %s | um %s' % tuple(
        ' '.join('%.0f' % t[i * q:(i + 1) * q].mean() for i in range(4)) for t in (hm, hm3, um)))
    # measured on MI355X (300 steps): hm 8912 -> 409, hm3 21354 -> 1589, um 65268 -> 21262
    assert hm[-10:].mean() < 0.1 * hm[:3].mean() and hm3[-10:].mean() < 0.15 * hm3[:3].mean() and um[-10:].mean() < 0.5 * um[:3].mean()
    for t in (hm, hm3, um):                                  # "monotonically-ish": every quarter of the run lower than the one before
        quarters = [t[i * q:(i + 1) * q].mean() for i in range(4)]
        assert quarters[0] > quarters[1] > quarters[2] > quarters[3], quarters
    held = trained['held']
    xyz = evaluate(trained['params'], S, F, DATASET, held)
    e = joint_error_mm(xyz, held[1])
    print('held-out synthetic hands (400 frames x %d joints): mean joint error %.2f mm after %d optimizer steps (untrained: %.1f mm), '
          'median %.2f, 99 %% %.1f' % (J, e.mean(), STEPS, trained['err0'], np.median(e), np.quantile(e, 0.99)))
    assert e.mean() * 10 < trained['err0'], (e.mean(), trained['err0'])


def _oracle_xyz(params, dm, cfgs, coms, dtype=None):
    import torch
    from oracle import net, pose
    from oracle.graph import NetConfig
    cfg = NetConfig(S, F, J)
    ndm = pose.norm_dm(dm, coms)
    ep = net.forward_eval(cfg, params, ndm, **({} if dtype is None else {'dtype': dtype}))
    maps = [np.asarray(ep[k][-1], np.float32) for k in ('hm_outs', 'hm3_outs', 'um_outs')]
    return pose.estimate_pose_mm(maps[0], maps[1], maps[2], ndm, cfgs, coms), maps


def test_trained_weights_one_engine_b40_all_joints_within_the_bar(trained):
    """BASELINE config 2's shape (ICVL S=2 F=128 B=40, forward(eval) + vote) on the trained weights: every head map within 5e-4 and
    EVERY joint of the batch against the oracle -- mean delta <= 0.1 mm, no joint further than 1 mm."""
    import torch
    from densereg_amd.engine import Engine
    from oracle import pose
    dm, gt, cfgs, coms = (a[:B] for a in trained['held'])
    ref, maps = _oracle_xyz(trained['params'], dm, cfgs, coms)
    eng = Engine(S, F, J, 128, 3, B, 0, training=False)
    eng.load_params(trained['params'])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    d_dm = eng.norm_dm(t(dm), t(coms))
    got_maps = [m.cpu().numpy() for m in eng.forward_eval(d_dm)]
    xyz = eng.infer(d_dm, t(cfgs), t(coms)).cpu().numpy()
    eng.close()
    for name, a, b in zip(('hm', 'hm3', 'um'), got_maps, maps):
        assert np.abs(a - b).max() < 5e-4, (name, float(np.abs(a - b).max()))
    d = np.linalg.norm((xyz - ref).reshape(-1, 3), axis=1)
    e_hip, e_ref = pose.mean_jnt_error(xyz, gt), pose.mean_jnt_error(ref, gt)
    print('one engine, B=40, trained weights: mean joint error %.4f mm (oracle %.4f, delta %.5f); per joint vs the oracle: mean %.2e mm, '
          'max %.2e mm over all %d joints' % (e_hip, e_ref, abs(e_hip - e_ref), d.mean(), d.max(), d.size))
    assert abs(e_hip - e_ref) <= 0.1 and d.mean() <= 0.1 and d.max() <= 1.0, (e_hip, e_ref, d.mean(), d.max())


def test_trained_weights_replica_pool_2x5_all_joints_within_the_bar(trained):
    """``bench.py``'s forward+vote leg as it is timed (``ReplicaPool(2, merge=5)``: launches of 200 crops) on the trained weights: all
    6400 joints of 400 held-out frames against the oracle, no exclusions."""
    import torch
    from densereg_amd.serving import ReplicaPool
    from oracle import pose
    dm, gt, cfgs, coms = trained['held']
    pool = ReplicaPool(2, S, F, J, 128, 3, B, 0, merge=5)
    pool.load_params(trained['params'])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(pool.device)
    tickets = []
    for i in range(0, dm.shape[0], B):
        sl = slice(i, i + B)
        tickets.append(pool.submit(pool.norm_dm(t(dm[sl]), t(coms[sl])), t(cfgs[sl]), t(coms[sl])))
    xs = []
    for xyz, ticket in tickets:
        pool.wait(ticket)
        torch.cuda.current_stream(pool.device).synchronize()
        xs.append(xyz.cpu().numpy())
    pool.close()
    a = np.concatenate(xs)
    ref = np.concatenate([_oracle_xyz(trained['params'], dm[i:i + B], cfgs[i:i + B], coms[i:i + B])[0] for i in range(0, dm.shape[0], B)])
    d = np.linalg.norm((a - ref).reshape(-1, 3), axis=1)
    e_hip, e_ref = pose.mean_jnt_error(a, gt), pose.mean_jnt_error(ref, gt)
    print('ReplicaPool(2, merge=5), trained weights, %d frames: mean joint error %.4f mm (oracle %.4f, delta %.5f); per joint vs the oracle: '
          'mean %.2e mm, max %.2e mm, %d of %d joints further than 0.1 mm' % (a.shape[0], e_hip, e_ref, abs(e_hip - e_ref), d.mean(), d.max(),
                                                                               int((d > 0.1).sum()), d.size))
    assert abs(e_hip - e_ref) <= 0.1 and d.mean() <= 0.1 and d.max() <= 1.0, (e_hip, e_ref, d.mean(), d.max())
