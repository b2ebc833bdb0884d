"""Parity of the HIP training path (train-mode BatchReNorm forward, loss, backward, clip+Adam) against
the CPU oracle's autograd, through the C ABI -- ``[emu]`` on CPU fibers, ``[gpu]`` on an MI355X.

Gradient tolerance: this network's fp32 gradients are noisy by construction (ReLU/max-pool switches and
the cancellation in BatchNorm's backward): the oracle's OWN fp32 autograd differs from its fp64 autograd
by up to ~1e-2..3e-1 of a tensor's max (measured; printed below).  So the engine is checked against the
fp64 oracle with (a) per-tensor max error <= 6e-2 of the tensor's max (or 1.25x torch-fp32's own worst tensor where
that is larger: the deep / wide configurations), (b) median <= 1e-2, and (c) not noisier than torch's fp32 autograd of
the same graph by more than 2x.
"""
import ctypes as C
import os

import numpy as np
import pytest

from tests.common import _flat_rw, flat_grads_by_name, grad_metrics

BACKENDS = [pytest.param('emu'), pytest.param('gpu', marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def be(request):
    return request.getfixturevalue(request.param)


# per-tensor relative-L2 / cosine bars of the fp32 gradient against the oracle's fp32 autograd (set from the MI355X measurement printed
# by _run_step; the max-norm bars stay what they were)
# Measured (MI355X, S=2 F=128, fp32 engine vs torch-fp32 autograd -- two independent fp32 noises): worst tensor 1.2e-2 .. 2.8e-2 (a
# BatchReNorm beta / gamma: sums with cancellation), median 4.4e-3 .. 6.7e-3, cosine >= 0.99964; against the fp64 autograd the engine and
# torch-fp32 are equally far (S=1 F=64: 1.5e-3 / 6.7e-4 vs 1.3e-3 / 6.3e-4) -- that relative statement is asserted wherever fp64 runs.
GRAD_L2_WORST, GRAD_L2_MEDIAN, GRAD_COS_MIN = 6e-2, 1.5e-2, 0.999


def _case(S, F, J, B, dataset='icvl'):
    from densereg_amd.data.synthetic import make_crops
    from oracle import net, pose
    from oracle.graph import NetConfig
    cfg = NetConfig(S, F, J)
    dm, poses, cfgs, coms, _ = make_crops(B, dataset)
    poses = np.ascontiguousarray(poses[:, :3 * J])
    ndm = pose.norm_dm(dm, coms)
    calib = pose.norm_dm(*[make_crops(4, dataset, seed=5)[i] for i in (0, 3)])
    params = net.make_test_params(cfg, calib)
    return cfg, params, ndm, poses, cfgs, coms


def _record_branch(branch, B, cfg):
    """Which gradient bar a training-step test applied (the fp64 autograd needs host RAM): appended to
    gpurun_out/test_branches.jsonl so that a GPU run says what it checked."""
    import json
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'test_branches.jsonl'), 'a') as f:
            f.write(json.dumps({'test': os.environ.get('PYTEST_CURRENT_TEST', ''), 'gradient_bar': branch, 'B': int(B),
                                'S': cfg.num_stack, 'F': cfg.num_fea, 'J': cfg.num_jnt}) + '\n')
    except OSError:
        pass


def _run_step(be, cfg, params, ndm, poses, cfgs, coms, masks, ref64=True):
    import torch
    from oracle import net, train
    B, S, J = ndm.shape[0], cfg.num_stack, cfg.num_jnt
    h = be.handle(cfg, B, training=True)
    h.load_params(params)
    h.call('dr_finalize_params', be.stream)
    d_dm = be.dev(ndm)
    if masks is None:
        h.call('dr_forward_train', B, be.ptr(d_dm), 0, None, C.c_uint64(0), be.stream)
    else:
        d_mask = be.dev(np.ascontiguousarray(np.stack(masks)))
        h.call('dr_forward_train', B, be.ptr(d_dm), 1, be.ptr(d_mask), C.c_uint64(0), be.stream)
    lo32, g32, upd, outs = train.loss_and_grads(cfg, params, ndm, poses, cfgs, coms, dropout_masks=masks)
    m = cfg.out_hw
    for s in range(S):
        hm, hm3, um = be.empty((B, m, m, J)), be.empty((B, m, m, J)), be.empty((B, m, m, 3 * J))
        h.call('dr_read_maps', B, s, be.ptr(hm), be.ptr(hm3), be.ptr(um), be.stream)
        be.sync()
        for got, key in ((hm, 'hm_outs'), (hm3, 'hm3_outs'), (um, 'um_outs')):
            dev = float(np.abs(be.host(got) - outs[key][s]).max())
            print('train-mode maps vs oracle fp32: stack %d %s max |diff| %.3g' % (s, key, dev))
            assert dev < 5e-4, (s, key, dev)
    d_pose, d_cfg, d_com, d_lo = be.dev(poses), be.dev(cfgs), be.dev(coms), be.empty((4,))
    h.call('dr_loss', B, be.ptr(d_dm), be.ptr(d_pose), be.ptr(d_cfg), be.ptr(d_com), be.ptr(d_lo), be.stream)
    be.sync()
    np.testing.assert_allclose(be.host(d_lo), [lo32[k] for k in ('hm', 'hm3', 'um', 'reg')], rtol=2e-4)
    h.call('dr_zero_grad', be.stream)
    h.call('dr_backward', B, be.stream)
    be.sync()
    g = flat_grads_by_name(be, h, cfg)
    # (1) the bar that does not depend on the host: engine vs the oracle's fp32 autograd, two fp32 noises of this network
    # (ReLU / max-pool switches, BatchNorm cancellation) against each other -- every tensor within 1.2e-1 of its max, median 2e-2
    e_pair = np.array([np.abs(g[n] - g32[n]).max() / (np.abs(g32[n]).max() + 1e-12) for n in g32])
    print('grad error vs the fp32 oracle: engine max %.2e median %.2e' % (e_pair.max(), np.median(e_pair)))
    # (asserted on the S <= 2 networks at 128x128, where the fp64 statement may be out of the host's reach at B=40; the deep S=4
    # F=256 network -- torch-fp32 itself is 0.46 from fp64 on its worst tensor -- is always small enough for (2))
    if cfg.num_stack <= 2 and cfg.in_hw == 128:
        assert e_pair.max() < 1.6e-1 and np.median(e_pair) < 2e-2, (e_pair.max(), np.median(e_pair))
    # (1b) the statement a max-norm cannot make: per tensor, relative L2 error and cosine against the oracle's gradient -- a
    # systematic few-per-cent error of one tensor (a wrong scale, a missing term) moves these, a single ReLU switch does not
    names, _, l2_pair, cos_pair = grad_metrics(g, g32)
    w = int(np.argmax(l2_pair))
    print('grad vs the fp32 oracle, per tensor: rel-L2 max %.2e (%s) median %.2e | cosine min %.6f median %.8f'
          % (l2_pair.max(), names[w], np.median(l2_pair), cos_pair.min(), np.median(cos_pair)))
    big = np.array([g32[n].size >= 4096 for n in names])
    if big.any():
        print('   ... the tensors of >= 4096 elements (conv weights): rel-L2 max %.2e median %.2e | cosine min %.6f'
              % (l2_pair[big].max(), np.median(l2_pair[big]), cos_pair[big].min()))
    if cfg.num_stack <= 2 and cfg.in_hw == 128:               # (the deep S=4 F=256 network: the fp64 statement below)
        assert l2_pair.max() < GRAD_L2_WORST and np.median(l2_pair) < GRAD_L2_MEDIAN and cos_pair.min() > GRAD_COS_MIN, \
            (l2_pair.max(), names[w], np.median(l2_pair), cos_pair.min())
    _record_branch('ref64' if ref64 else 'fp32-only', B, cfg)
    g64 = g32
    if ref64:
        # (2) where the host has the memory for the oracle's fp64 autograd (~0.75 GB per crop at S=2 F=128): the engine is no
        # noisier than torch's own fp32 autograd of the same graph, both measured against fp64
        _, g64, _, _ = train.loss_and_grads(cfg, params, ndm, poses, cfgs, coms, dropout_masks=masks, dtype=torch.float64)
        e_eng, e_o32 = [], []
        for name, ref in g64.items():
            sc = np.abs(ref).max() + 1e-12
            e_eng.append(np.abs(g[name] - ref).max() / sc)
            e_o32.append(np.abs(g32[name] - ref).max() / sc)
        e_eng, e_o32 = np.array(e_eng), np.array(e_o32)
        print('grad error vs fp64 oracle: engine max %.2e median %.2e | torch-fp32 max %.2e median %.2e'
              % (e_eng.max(), np.median(e_eng), e_o32.max(), np.median(e_o32)))
        # (a) worst tensor: 6e-2 of its max, or -- on the deep / wide configurations, where torch's own fp32 autograd is
        # already above that (measured on MI355X: config 3 at B=40 torch 7.65e-2 / engine 7.69e-2; S=4 F=256 at 256x256
        # torch 0.46 / engine 0.33) -- no worse than 1.25x torch-fp32's worst tensor; (b), (c) medians
        assert e_eng.max() < max(6e-2, 1.25 * e_o32.max()), (e_eng.max(), e_o32.max())
        assert np.median(e_eng) < max(1e-2, 1.25 * np.median(e_o32))
        assert np.median(e_eng) < 2 * np.median(e_o32) + 1e-4
        _, _, l2_e, cos_e = grad_metrics(g, g64)
        _, _, l2_o, cos_o = grad_metrics(g32, g64)
        print('grad vs the fp64 oracle, per tensor rel-L2: engine max %.2e median %.2e | torch-fp32 max %.2e median %.2e ; cosine min: engine %.6f torch-fp32 %.6f'
              % (l2_e.max(), np.median(l2_e), l2_o.max(), np.median(l2_o), cos_e.min(), cos_o.min()))
        # the engine's fp32 gradient is as close to the fp64 gradient as torch's fp32 autograd of the same graph is
        assert l2_e.max() < max(2 * l2_o.max(), 1e-3) and np.median(l2_e) < max(2 * np.median(l2_o), 1e-4), (l2_e.max(), l2_o.max())
        assert 1 - cos_e.min() < max(4 * (1 - cos_o.min()), 1e-6), (cos_e.min(), cos_o.min())
    # a second loss+backward on the same forward must add exactly the same gradient again: catches gradient
    # buffers that are neither zeroed nor overwritten by their first writer (train_exec.inc plan_backward)
    h.call('dr_loss', B, be.ptr(d_dm), be.ptr(d_pose), be.ptr(d_cfg), be.ptr(d_com), be.ptr(d_lo), be.stream)
    h.call('dr_backward', B, be.stream)
    be.sync()
    g2 = flat_grads_by_name(be, h, cfg)
    for name in g:
        sc = np.abs(g[name]).max() + 1e-12
        assert np.abs(g2[name] - 2.0 * g[name]).max() / sc < 1e-4, name
    assert h.lib.dr_lookback_expired(h._h) == 0          # no look-back wait of the BatchReNorm hand-off ever ran out
    # BatchReNorm state after one micro-step (moving stats with zero-debias, r_max/d_max/curr_t schedule)
    p2 = {k: v.copy() for k, v in params.items()}
    net.bn_state_update(p2, upd, zero_debias=True, shadow={})
    got = h.read_params()
    for k in p2:
        if 'moving' in k or k.endswith(('r_max', 'd_max', 'curr_t')):
            # atol scales with the tensor: on the deep S=4 net the batch means reach ~100 and an element near zero carries the
            # fp32 rounding of its neighbours' magnitude
            np.testing.assert_allclose(got[k], p2[k], rtol=2e-4, atol=5e-5 * max(1.0, float(np.abs(p2[k]).max())), err_msg=k)
    return h, g64


def test_train_step_single_stack(be):
    cfg, params, ndm, poses, cfgs, coms = _case(1, 64, 4, 1 if be.name == 'emu' else 3)
    h, _ = _run_step(be, cfg, params, ndm, poses, cfgs, coms, None)
    h.close()


def test_train_step_input_256(be):
    """BASELINE config 5's geometry in training: 256x256 crops, hourglass depth 5, 64x64 maps (um_v1.py:99-104) -- one
    micro-step of a narrow network (S=1, F=16, J=3) against the oracle: losses and every gradient."""
    import torch
    from densereg_amd.data.synthetic import make_crops
    from oracle import net, pose, train
    from oracle.graph import NetConfig
    cfg = NetConfig(1, 16, 3, in_hw=256)
    dm, poses, cfgs, coms, _ = make_crops(1, 'nyu', seed=3, hw=256)
    poses = np.ascontiguousarray(poses[:, :9])
    ndm = pose.norm_dm(dm, coms)
    params = net.make_test_params(cfg, ndm, seed=7)
    h = be.handle(cfg, 1, training=True)
    h.load_params(params)
    h.call('dr_finalize_params', be.stream)
    d_dm, d_pose, d_cfg, d_com, d_lo = be.dev(ndm), be.dev(poses), be.dev(cfgs), be.dev(coms), be.empty((4,))
    h.call('dr_forward_train', 1, be.ptr(d_dm), 0, None, C.c_uint64(0), be.stream)
    h.call('dr_loss', 1, be.ptr(d_dm), be.ptr(d_pose), be.ptr(d_cfg), be.ptr(d_com), be.ptr(d_lo), be.stream)
    h.call('dr_zero_grad', be.stream)
    h.call('dr_backward', 1, be.stream)
    be.sync()
    g = flat_grads_by_name(be, h, cfg)
    lo32, g32, _, _ = train.loss_and_grads(cfg, params, ndm, poses, cfgs, coms)
    _, g64, _, _ = train.loss_and_grads(cfg, params, ndm, poses, cfgs, coms, dtype=torch.float64)
    np.testing.assert_allclose(be.host(d_lo), [lo32[k] for k in ('hm', 'hm3', 'um', 'reg')], rtol=2e-4)
    e_eng = np.array([np.abs(g[n] - g64[n]).max() / (np.abs(g64[n]).max() + 1e-12) for n in g64])
    e_o32 = np.array([np.abs(g32[n] - g64[n]).max() / (np.abs(g64[n]).max() + 1e-12) for n in g64])
    assert e_eng.max() < 6e-2 and np.median(e_eng) < 2 * np.median(e_o32) + 1e-4, (e_eng.max(), np.median(e_eng), np.median(e_o32))
    h.close()


def _bf16_step_check(be, cfg, params, ndm, poses, cfgs, coms, med=1.15, q90=1.4, worst=2.5):
    """One training micro-step on the bf16 matrix cores against the oracle's bf16-operand evaluation, both measured against the
    fp64 oracle (relative L2 per gradient tensor): the engine carries the precision's own noise and nothing else."""
    import torch
    from oracle import train
    B = ndm.shape[0]
    h = be.handle(cfg, B, training=True)
    h.call('dr_set_precision', 1)
    h.load_params(params)
    h.call('dr_finalize_params', be.stream)
    d_dm, d_pose, d_cfg, d_com, d_lo = be.dev(ndm), be.dev(poses), be.dev(cfgs), be.dev(coms), be.empty((4,))
    h.call('dr_forward_train', B, be.ptr(d_dm), 0, None, C.c_uint64(0), be.stream)
    h.call('dr_loss', B, be.ptr(d_dm), be.ptr(d_pose), be.ptr(d_cfg), be.ptr(d_com), be.ptr(d_lo), be.stream)
    h.call('dr_zero_grad', be.stream)
    h.call('dr_backward', B, be.stream)
    be.sync()
    g = flat_grads_by_name(be, h, cfg)
    lo16, g16, _, _ = train.loss_and_grads(cfg, params, ndm, poses, cfgs, coms, conv_operands='bf16')
    _, g64, _, _ = train.loss_and_grads(cfg, params, ndm, poses, cfgs, coms, dtype=torch.float64)
    np.testing.assert_allclose(be.host(d_lo), [lo16[k] for k in ('hm', 'hm3', 'um', 'reg')], rtol=1e-2)
    l2 = lambda a, b: float(np.linalg.norm((a - b).ravel()) / (np.linalg.norm(np.asarray(b).ravel()) + 1e-30))
    e_eng = np.array([l2(g[n], g64[n]) for n in g64])
    e_prec = np.array([l2(g16[n], g64[n]) for n in g64])
    ratio = e_eng / (e_prec + 1e-12)
    print('bf16 gradient error vs fp64 oracle (relative L2 per tensor): engine median %.2e max %.2e | oracle-bf16 median %.2e max %.2e | '
          'ratio median %.2f q90 %.2f max %.2f' % (np.median(e_eng), e_eng.max(), np.median(e_prec), e_prec.max(),
                                                   np.median(ratio), float(np.quantile(ratio, 0.9)), float(ratio.max())))
    assert np.isfinite(e_eng).all()
    assert np.median(e_eng) <= med * np.median(e_prec) + 1e-6
    assert np.quantile(ratio, 0.9) <= q90 and (e_eng <= worst * e_prec + 1e-4).all(), (float(np.quantile(ratio, 0.9)), float(ratio.max()))
    assert np.median(e_prec) > 1e-3                       # the comparison is about bf16, not fp32
    return h, (d_dm, d_pose, d_cfg, d_com, d_lo)


def test_train_step_bf16_precision(be):
    """dr_set_precision(DR_PREC_BF16) on a training handle: forward, input-gradient and weight-gradient convolutions
    all round their two operands to bf16 on the way into the matrix cores (fp32 accumulation, fp32 tensors, fp32
    BatchReNorm / loss / Adam).  The oracle's statement of that arithmetic is oracle/net.py::_ConvBf16Operands.  On this
    network bf16 gradients are far noisier than fp32 ones (ReLU / max-pool switches and BatchNorm cancellation amplify
    a 2^-9 operand rounding: the ORACLE's bf16 gradients differ from its fp64 gradients by ~0.4 relative L2 per tensor
    at B=1, its fp32 gradients by 2e-3), so the criterion is the one of the forward test: the engine carries the
    precision's own noise and nothing else -- its distance to fp64 is within 1.15x of the oracle's bf16 evaluation in the
    median over tensors, within 1.4x for 90 % of them and 2.5x for every one (measured on MI355X, B=3: median ratio
    1.05, worst tensor 1.73); losses within 1 % of the oracle's bf16 evaluation."""
    cfg, params, ndm, poses, cfgs, coms = _case(1, 64, 4, 1 if be.name == 'emu' else 3)
    B = ndm.shape[0]
    h, (d_dm, d_pose, d_cfg, d_com, d_lo) = _bf16_step_check(be, cfg, params, ndm, poses, cfgs, coms)
    # one optimizer step runs (weights re-packed as bf16 for both conv directions) and the next forward is finite
    h.call('dr_apply_adam', C.c_float(1e-3), C.c_float(1.0), C.c_float(0.2), C.c_int64(1), be.stream)
    h.call('dr_forward_train', B, be.ptr(d_dm), 0, None, C.c_uint64(0), be.stream)
    h.call('dr_loss', B, be.ptr(d_dm), be.ptr(d_pose), be.ptr(d_cfg), be.ptr(d_com), be.ptr(d_lo), be.stream)
    be.sync()
    assert np.isfinite(be.host(d_lo)).all()
    h.close()


def test_train_step_bf16_precision_with_fp32_storage(be, monkeypatch):
    """The same criterion with every bf16 STORAGE decision of the training step switched off (DR_BF16_ACT / DRAW / RAW / GACT = 0:
    single-reader activations and their gradients, dRaw and the raw outputs of BatchReNorm convs all fp32 in HBM) -- the paths the
    defaults no longer take: the bf16 conv kernels' third epilogue copy reading an fp32 raw output, the fp32-raw BatchReNorm
    passes beside bf16 matrix cores.  The switches are read when the handle is created."""
    for k in ('DR_BF16_ACT', 'DR_BF16_DRAW', 'DR_BF16_RAW', 'DR_BF16_GACT'):
        monkeypatch.setenv(k, '0')
    cfg, params, ndm, poses, cfgs, coms = _case(1, 64, 4, 1 if be.name == 'emu' else 3)
    h, _ = _bf16_step_check(be, cfg, params, ndm, poses, cfgs, coms)
    h.close()


def test_train_step_bf16_default_width_runs(be):
    """Regression (round-2 advisor finding): bf16 training at the default width F=128 with a per-rank batch of 3 failed in
    dr_backward with 'dgrad Conv_..: unsupported layout' -- the bf16-dRaw storage decision predicted the dgrad tile from all
    NpT columns while the launch narrows a concat slice with trailing uvd planes (131 -> 128 columns) and landed on the
    split-K kernel, which stages fp32 only.  The prediction now uses the launch's own column count (and a bf16-stored operand
    never selects split-K): one micro-step must run and give finite losses and gradients at F=128, B=3."""
    cfg, params, ndm, poses, cfgs, coms = _case(1, 128, 4, 3)
    B = ndm.shape[0]
    h = be.handle(cfg, B, training=True)
    h.call('dr_set_precision', 1)
    h.load_params(params)
    h.call('dr_finalize_params', be.stream)
    d_dm, d_pose, d_cfg, d_com, d_lo = be.dev(ndm), be.dev(poses), be.dev(cfgs), be.dev(coms), be.empty((4,))
    h.call('dr_forward_train', B, be.ptr(d_dm), 0, None, C.c_uint64(0), be.stream)
    h.call('dr_loss', B, be.ptr(d_dm), be.ptr(d_pose), be.ptr(d_cfg), be.ptr(d_com), be.ptr(d_lo), be.stream)
    h.call('dr_zero_grad', be.stream)
    h.call('dr_backward', B, be.stream)
    be.sync()
    assert np.isfinite(be.host(d_lo)).all()
    g = flat_grads_by_name(be, h, cfg)
    assert all(np.isfinite(v).all() for v in g.values())
    assert sum(float(np.abs(v).sum()) for v in g.values()) > 0
    h.close()


def test_read_activation_of_a_bf16_stored_tensor(be, monkeypatch):
    """On the bf16 path the training forward stores a single-conv-reader activation as bf16 (DR_BF16_ACT); dr_read_activation
    widens it instead of refusing (round-2 advisor finding): the values are the nearest-even bf16 of what a handle with
    DR_BF16_ACT=0 holds in fp32."""
    from oracle.graph import conv_specs
    from tests.common import bf16_round
    B = 3                                  # 12 288 pixels at 64x64: the reader runs on a tiled kernel, not the fp32-staging split-K one
    cfg, params, ndm, poses, cfgs, coms = _case(1, 64, 4, B)
    specs = conv_specs(cfg)
    # the first conv of the stem's first residual module (1x1 32 -> 16 at 64x64): BatchReNorm + ReLU, read only by the module's 3x3 conv
    name = [c.name for c in specs if c.bn and c.k == 1 and c.h_out == 64 and c.cout == 16][0]
    shape = (B, 64, 64, 16)

    def act(stored):
        monkeypatch.setenv('DR_BF16_ACT', '1' if stored else '0')
        h = be.handle(cfg, B, training=True)
        h.call('dr_set_precision', 1)
        h.load_params(params)
        h.call('dr_finalize_params', be.stream)
        d_dm = be.dev(ndm)
        h.call('dr_forward_train', B, be.ptr(d_dm), 0, None, C.c_uint64(0), be.stream)
        a = be.read_activation(h, name, shape)
        h.close()
        return a
    a32, a16 = act(False), act(True)
    assert np.isfinite(a16).all() and np.abs(a32).max() > 0
    np.testing.assert_array_equal(a16, bf16_round(a32))


def test_grouped_small_layer_wgrad_matches_per_layer_launches(be, monkeypatch):
    """Layers up to 16x16 pixels keep their dRaw in a private buffer and compute their weight gradients in ONE grouped
    launch at the end of the backward sweep (conv_wgrad_group_kernel; DR_GROUP_WGRAD=0 restores a launch per layer).
    Same kernel body, other slab cuts: every gradient must agree to fp32 summation-order noise, across two micro-steps
    (the private buffers and the group table are reused) and a second batch size (a new table)."""
    cfg, params, ndm, poses, cfgs, coms = _case(1, 64, 4, 2 if be.name == 'emu' else 5)

    def run(group):
        monkeypatch.setenv('DR_GROUP_WGRAD', '1' if group else '0')
        B = ndm.shape[0]
        h = be.handle(cfg, B, training=True)
        h.load_params(params)
        h.call('dr_finalize_params', be.stream)
        out = []
        for Bn in ((B, B - 1) if be.name == 'emu' else (B, B, B - 1)):       # (emulator: CPU time)
            d_dm, d_pose, d_cfg, d_com, d_lo = (be.dev(np.ascontiguousarray(a[:Bn])) for a in (ndm, poses, cfgs, coms, np.zeros((B, 4), np.float32)))
            h.call('dr_forward_train', Bn, be.ptr(d_dm), 0, None, C.c_uint64(0), be.stream)
            h.call('dr_loss', Bn, be.ptr(d_dm), be.ptr(d_pose), be.ptr(d_cfg), be.ptr(d_com), be.ptr(d_lo), be.stream)
            h.call('dr_zero_grad', be.stream)
            h.call('dr_backward', Bn, be.stream)
            be.sync()
            out.append(flat_grads_by_name(be, h, cfg))
        h.close()
        return out
    grouped, single = run(True), run(False)
    # the full-resolution layers' weight gradients on the library's low-priority side stream, released when the sweep enters
    # an hourglass (the default; DR_WGRAD_STREAM=0 = inline): same kernels, same slab plan
    monkeypatch.setenv('DR_WGRAD_STREAM', '0')            # the default is on: compare with everything inline on one stream
    inline = run(True)
    monkeypatch.delenv('DR_WGRAD_STREAM')
    for ga, gb in zip(inline, grouped):
        for n in ga:
            assert np.abs(ga[n] - gb[n]).max() / (np.abs(gb[n]).max() + 1e-12) < 2e-5, n
    if be.name == 'gpu':                                  # (the hand-off's kernels are covered on the emulator by test_bn_layer.py)
        monkeypatch.setenv('DR_BN_LOOKBACK', '1')         # opt-in (measured slower on MI355X), kept correct
        lookback = run(True)
        monkeypatch.delenv('DR_BN_LOOKBACK')
        for ga, gb in zip(lookback, grouped):
            for n in ga:
                assert np.abs(ga[n] - gb[n]).max() / (np.abs(gb[n]).max() + 1e-12) < 2e-5, n
    differs = 0
    for ga, gb in zip(grouped, single):
        for n in ga:
            sc = np.abs(gb[n]).max() + 1e-12
            assert np.abs(ga[n] - gb[n]).max() / sc < 2e-5, n
            differs += int(np.abs(ga[n] - gb[n]).max() > 0)
    # other slab cuts = another summation order: the grouped path really ran (with executor lanes on, DR_MULTI_STREAM=1, the
    # grouped launch is off by design and the two runs are bit-identical: the step has no floating-point atomics)
    assert differs > 0 or os.environ.get('DR_MULTI_STREAM') == '1'


def test_bf16_draw_storage_is_numerically_transparent(be):
    """On the bf16 matrix-core path the gradient wrt a BatchReNorm layer's raw output is STORED as bf16 (half the bytes in
    three passes): its only readers, the layer's input-gradient conv and its weight gradient, round it to bf16 while staging
    anyway.  Checked kernel by kernel, bit for bit (the whole-network gradients cannot say it: two identical bf16 runs already
    differ by ~1e-3 through the order of the fp atomics in the stem moments): (1) the BatchReNorm backward apply writes exactly
    the nearest-even bf16 of what it writes in fp32; (2) the conv kernels, every tile incl. ragged channel counts whose last
    16-byte slot hangs over the row, and (3) the weight-gradient kernel give identical results from bf16-stored and
    fp32-stored operands of the same values."""
    import ctypes as Cc
    import torch
    from densereg_amd import _lib
    from tests.common import bf16_round
    from tests.test_bn_layer import _run as bn_run
    rng = np.random.default_rng(5)

    def to_bf16_bits(a):                                   # fp32 array of bf16-representable values -> uint16 bit patterns
        return (np.ascontiguousarray(a, np.float32).view(np.uint32) >> 16).astype(np.uint16)
    lib = be.dbg
    try:
        assert lib.dr_dbg_force_bf16(1) == 0
        # ---- (2) conv: x stored as bf16 vs the same values stored as fp32
        for tile in (1, 3, 4, 7, 8):
            for (cin, cout, k, hw) in ((36, 61, 3, (6, 5)), (65, 93, 1, (4, 7)), (16, 30, 3, (5, 5)), (140, 150, 1, (3, 4))):
                B = 2
                x = bf16_round(rng.standard_normal((B,) + hw + (cin,)).astype(np.float32))
                w = (rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
                x_cs = -(-cin // 4) * 4
                xp = np.zeros((B,) + hw + (x_cs,), np.float32)      # channel groups of four are zero-padded by the producer
                xp[..., :cin] = x
                outs = []
                for stored in (0, 1):
                    assert lib.dr_dbg_force_bf16_storage(stored) == 0 and lib.dr_dbg_force_tile(tile) == 0
                    xin = np.concatenate([to_bf16_bits(xp).reshape(-1), np.full(64, 0x7FC0, np.uint16)]) if stored else xp   # NaN after the tensor
                    d_x, d_w = be.dev(xin), be.dev(w)
                    d_y = be.dev(np.zeros((B,) + hw + (cout,), np.float32))
                    rc = lib.dr_dbg_conv2d(B, hw[0], hw[1], cin, cout, k, be.ptr(d_x), x_cs, be.ptr(d_w), None, None, 0, None, 0, None, 0.0,
                                           be.ptr(d_y), cout, None, be.stream)
                    assert rc == 0, rc
                    be.sync()
                    outs.append(be.host(d_y).copy())
                np.testing.assert_array_equal(outs[0], outs[1], err_msg='conv tile %d %s' % (tile, (cin, cout, k)))
        lib.dr_dbg_force_tile(-1)
        # ---- (3) weight gradient: g stored as bf16
        for (cin, cout, k, hw, T) in ((40, 61, 3, (6, 6), 64), (130, 70, 1, (5, 8), 128), (64, 36, 3, (4, 4), 64)):
            B = 2
            x = bf16_round(rng.standard_normal((B,) + hw + (cin,)).astype(np.float32))
            g = bf16_round(rng.standard_normal((B,) + hw + (cout,)).astype(np.float32))
            x_cs, g_cs = -(-cin // 4) * 4, -(-cout // 4) * 4
            xp = np.zeros((B,) + hw + (x_cs,), np.float32); xp[..., :cin] = x
            gp = np.zeros((B,) + hw + (g_cs,), np.float32); gp[..., :cout] = g
            outs = []
            for stored in (0, 1):
                assert lib.dr_dbg_force_bf16_storage(stored) == 0
                gin = np.concatenate([to_bf16_bits(gp).reshape(-1), np.full(64, 0x7FC0, np.uint16)]) if stored else gp
                xin = np.concatenate([to_bf16_bits(xp).reshape(-1), np.full(64, 0x7FC0, np.uint16)]) if stored else xp
                d_x, d_g = be.dev(xin), be.dev(gin)
                d_w = be.dev(np.zeros((k, k, cin, cout), np.float32))
                rc = lib.dr_dbg_wgrad(B, hw[0], hw[1], cin, cout, k, be.ptr(d_x), x_cs, be.ptr(d_g), g_cs, None, 0.0, T, 3, be.ptr(d_w), be.stream)
                assert rc == 0, rc
                be.sync()
                outs.append(be.host(d_w).copy())
            np.testing.assert_array_equal(outs[0], outs[1], err_msg='wgrad %s' % ((cin, cout, k),))
    finally:
        lib.dr_dbg_force_bf16_storage(0)
        lib.dr_dbg_force_bf16(0)
        lib.dr_dbg_force_tile(-1)
    # ---- (1) BatchReNorm apply passes: bf16 draw == RNE(fp32 draw), bf16 activation == RNE(fp32 activation), pads zero
    for case in ((2, 4, 4, 19, 65, 1), (8, 32, 32, 8, 78, 1)):
        bufs = []
        for stored in (0, 1):
            lib.dr_dbg_force_bf16_storage(stored)
            try:
                bufs.append(bn_run(be, *case, relu=True, with_res=False, seed=9, return_raw_draw=True))
            finally:
                lib.dr_dbg_force_bf16_storage(0)
        for which in (0, 1):                                # 0 = draw (backward apply), 1 = y (forward apply)
            f32, b16 = bufs[0][which], bufs[1][which]
            M, cs, Cout = f32.shape[0], f32.shape[1], case[4]
            bits = np.ascontiguousarray(b16).view(np.uint16).reshape(-1)[:M * cs].reshape(M, cs)
            got = (bits.astype(np.uint32) << 16).view(np.float32)
            np.testing.assert_array_equal(got[:, :Cout], bf16_round(f32[:, :Cout]))
            assert (got[:, Cout:] == 0).all()              # a reader's 16-byte slot may cover them


@pytest.mark.gpu
def test_train_step_config3_nyu_two_stacks_dropout_mask(gpu):
    """BASELINE.json config 3 shape (NYU S=2 F=128 J=14) at B=4 with an injected dropout mask."""
    cfg, params, ndm, poses, cfgs, coms = _case(2, 128, 14, 4, 'nyu')
    rng = np.random.default_rng(0)
    masks = [rng.integers(0, 2, (4, 32, 32, 512)).astype(np.uint8) for _ in range(4)]
    h, _ = _run_step(gpu, cfg, params, ndm, poses, cfgs, coms, masks)
    h.close()


@pytest.mark.gpu
def test_lanes_match_single_stream(gpu, monkeypatch):
    """The executor runs the two branches of every hourglass level on separate HIP streams (net.h: lanes).
    Five training micro-steps + an eval forward with lanes on (DR_MULTI_STREAM=1) must reproduce the default single-stream handle: maps
    bit-exact in eval, gradients to fp32 summation-order noise (lanes run the weight gradients inline with their own slab plans,
    the default puts them on the side stream / in the grouped launch).  A missing event edge shows up as a gross mismatch in
    some repetition."""
    cfg, params, ndm, poses, cfgs, coms = _case(2, 64, 8, 6, 'nyu')
    B = ndm.shape[0]

    def run(single):
        if single:
            monkeypatch.delenv('DR_MULTI_STREAM', raising=False)
        else:
            monkeypatch.setenv('DR_MULTI_STREAM', '1')
        h = gpu.handle(cfg, B, training=True)
        h.load_params(params)
        h.call('dr_finalize_params', gpu.stream)
        d_dm, d_pose, d_cfg, d_com, d_lo = gpu.dev(ndm), gpu.dev(poses), gpu.dev(cfgs), gpu.dev(coms), gpu.empty((4,))
        grads = []
        for _ in range(5):
            h.call('dr_forward_train', B, gpu.ptr(d_dm), 0, None, C.c_uint64(0), gpu.stream)
            h.call('dr_loss', B, gpu.ptr(d_dm), gpu.ptr(d_pose), gpu.ptr(d_cfg), gpu.ptr(d_com), gpu.ptr(d_lo), gpu.stream)
            h.call('dr_zero_grad', gpu.stream)
            h.call('dr_backward', B, gpu.stream)
            gpu.sync()
            grads.append(flat_grads_by_name(gpu, h, cfg))
        h.call('dr_set_fusion', 0)                         # (with lanes the hourglass bottoms are never fused: same launches on both handles)
        maps = gpu.forward_eval(h, ndm)
        h.close()
        return grads, maps

    g_multi, m_multi = run(False)
    g_single, m_single = run(True)
    for a, b in zip(m_multi, m_single):
        np.testing.assert_array_equal(a, b)
    for step, (ga, gb) in enumerate(zip(g_multi, g_single)):
        for name in ga:
            sc = np.abs(gb[name]).max() + 1e-12
            assert np.abs(ga[name] - gb[name]).max() / sc < 1e-5, (step, name)


def test_micro_step_is_bit_reproducible(be):
    """forward(train) + loss + backward twice on fresh handles: identical bits.  The path has no floating-point atomics --
    BatchReNorm sums go through per-workgroup partial rows folded in a fixed order (the stem's moments_kernel included), the
    max-pool backward gathers over the arg-max its forward recorded, weight-gradient slabs are folded in slab order."""
    cfg, params, ndm, poses, cfgs, coms = _case(2, 32, 5, 1 if be.name == 'emu' else 7)
    B = ndm.shape[0]

    def run():
        h = be.handle(cfg, B, training=True)
        h.load_params(params)
        h.call('dr_finalize_params', be.stream)
        d_dm, d_pose, d_cfg, d_com, d_lo = be.dev(ndm), be.dev(poses), be.dev(cfgs), be.dev(coms), be.empty((4,))
        h.call('dr_forward_train', B, be.ptr(d_dm), 2, None, C.c_uint64(11), be.stream)
        h.call('dr_loss', B, be.ptr(d_dm), be.ptr(d_pose), be.ptr(d_cfg), be.ptr(d_com), be.ptr(d_lo), be.stream)
        h.call('dr_zero_grad', be.stream)
        h.call('dr_backward', B, be.stream)
        be.sync()
        out = be.host(d_lo).copy(), flat_grads_by_name(be, h, cfg)
        h.close()
        return out

    (lo_a, g_a), (lo_b, g_b) = run(), run()
    np.testing.assert_array_equal(lo_a, lo_b)
    for name in g_a:
        np.testing.assert_array_equal(g_a[name], g_b[name], err_msg=name)


def test_bn_backward_sums_fused_into_dgrad(be, monkeypatch):
    """A BatchReNorm layer whose output has a single reader gets its backward sums from that reader's dgrad epilogue
    (conv_igemm.h bst_*).  DR_FUSE_BN_BWD=0 runs the separate reduce pass everywhere: same gradients to fp32/fp64
    summation-order noise."""
    cfg, params, ndm, poses, cfgs, coms = _case(1, 32, 4, 1 if be.name == 'emu' else 5)
    B = ndm.shape[0]

    def grads(fused):
        if fused:
            monkeypatch.delenv('DR_FUSE_BN_BWD', raising=False)
        else:
            monkeypatch.setenv('DR_FUSE_BN_BWD', '0')
        h = be.handle(cfg, B, training=True)
        h.load_params(params)
        h.call('dr_finalize_params', be.stream)
        d_dm, d_pose, d_cfg, d_com, d_lo = be.dev(ndm), be.dev(poses), be.dev(cfgs), be.dev(coms), be.empty((4,))
        h.call('dr_forward_train', B, be.ptr(d_dm), 0, None, C.c_uint64(0), be.stream)
        h.call('dr_loss', B, be.ptr(d_dm), be.ptr(d_pose), be.ptr(d_cfg), be.ptr(d_com), be.ptr(d_lo), be.stream)
        h.call('dr_zero_grad', be.stream)
        h.call('dr_backward', B, be.stream)
        be.sync()
        g = flat_grads_by_name(be, h, cfg)
        h.close()
        return g

    a, b = grads(True), grads(False)
    worst = max(float(np.abs(a[k] - b[k]).max() / (np.abs(b[k]).max() + 1e-12)) for k in a)
    assert worst < 2e-4, worst


WGRAD_CASES = [
    # B, H, W, Cin, Cout, k, T, nsplit, masked
    (2, 8, 8, 64, 64, 3, 64, 3, False),
    (1, 16, 16, 128, 128, 1, 128, 4, False),
    (2, 8, 8, 130, 70, 3, 128, 2, True),        # ragged channels on both sides, two ci tiles, row mask
    (1, 9, 7, 37, 45, 3, 64, 2, False),         # non power-of-two image: the division path of the loader
    (3, 4, 4, 200, 131, 1, 128, 1, True),
    (1, 2, 2, 64, 64, 3, 64, 1, False),         # image smaller than one 16-pixel step
    (2, 16, 16, 64, 70, 3, 64, 8, False),       # slabs in multiples of 8: the XCD-aware workgroup mapping
    (1, 16, 16, 130, 128, 1, 128, 16, True),
    (2, 8, 8, 78, 78, 3, 96, 2, False),         # kernel-row variant: one workgroup = the three taps of a row
    (1, 16, 16, 65, 96, 3, 96, 8, True),
    (2, 9, 7, 80, 70, 3, 96, 3, False),
    (1, 2, 2, 96, 65, 3, 96, 1, False),
    # 16x16-tile kernels (conv_wgrad16.h), T = 160 + id: 1 kernel row 80x80, 2 kernel row 32x32, 3 <= 80 inputs x 128-blocks of
    # outputs, 4 160-blocks of inputs x <= 80 outputs, 5 64x64 blocks
    (2, 8, 8, 78, 78, 3, 161, 2, False),
    (1, 16, 16, 65, 80, 3, 161, 8, True),
    (2, 9, 7, 70, 66, 3, 161, 3, False),
    (1, 2, 2, 80, 65, 3, 161, 1, False),
    (1, 8, 8, 100, 90, 3, 161, 2, False),       # more than one channel block on both sides
    (2, 16, 16, 16, 16, 3, 162, 4, False),
    (1, 9, 7, 32, 19, 3, 162, 2, True),
    (1, 8, 8, 40, 33, 3, 162, 1, False),
    (2, 8, 8, 78, 256, 1, 163, 3, False),
    (1, 16, 16, 65, 128, 1, 163, 8, True),
    (1, 5, 7, 70, 131, 1, 163, 2, False),
    (1, 4, 4, 90, 40, 1, 163, 1, False),
    (2, 8, 8, 156, 78, 1, 164, 2, False),
    (1, 16, 16, 131, 65, 1, 164, 16, True),
    (1, 6, 5, 170, 85, 1, 164, 3, False),
    (2, 16, 16, 32, 16, 1, 165, 4, False),
    (1, 8, 8, 16, 64, 1, 165, 2, True),
    (1, 9, 7, 32, 128, 1, 165, 3, False),
    (1, 4, 4, 70, 14, 1, 165, 1, False),
]


@pytest.mark.parametrize('case', WGRAD_CASES, ids=lambda c: 'x'.join(str(int(v)) for v in c))
def test_wgrad_kernel_direct(be, case):
    """conv_wgrad_kernel (both channel tiles) against an fp64 einsum of the definition."""
    B, H, W, Cin, Cout, k, T, nsplit, masked = case
    rng = np.random.default_rng(sum(case[:6]))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    g = rng.standard_normal((B, H, W, Cout)).astype(np.float32)
    mask = rng.uniform(-1, 1, B * H * W).astype(np.float32) if masked else None
    dw = be.wgrad(x, g, k, T, nsplit, mask, -0.25)
    xz = x.astype(np.float64)
    if masked:
        xz = xz * (~(mask.reshape(B, H, W, 1) < -0.25))
    pad = k // 2
    xp = np.pad(xz, ((0, 0), (pad, pad), (pad, pad), (0, 0)))
    ref = np.zeros((k, k, Cin, Cout))
    for dy in range(k):
        for dx in range(k):
            ref[dy, dx] = np.einsum('bhwc,bhwd->cd', xp[:, dy:dy + H, dx:dx + W], g.astype(np.float64))
    assert np.abs(dw - ref).max() / np.abs(ref).max() < 2e-5


@pytest.mark.parametrize('case', [c for c in WGRAD_CASES if c[6] in (64, 128)], ids=lambda c: 'x'.join(str(int(v)) for v in c))
def test_wgrad_x3_against_the_fp32_matrix_cores(be, case):
    """conv_wgrad_x3.h (both operands split into three bf16 planes while staged, six plane products on the bf16 matrix cores, the
    transpose by ds_read_b64_tr_b16) and conv_wgrad_kernel (fp32 matrix cores) on the same problem, both against the fp64 einsum:
    the same 2e-5 bar, and operands spread over many binades so that every plane carries weight."""
    B, H, W, Cin, Cout, k, T, nsplit, masked = case
    rng = np.random.default_rng(sum(case[:6]) + 77)
    x = (rng.standard_normal((B, H, W, Cin)) * np.exp(rng.uniform(-5, 5, (B, H, W, Cin)))).astype(np.float32)
    g = (rng.standard_normal((B, H, W, Cout)) * np.exp(rng.uniform(-5, 5, (B, H, W, Cout)))).astype(np.float32)
    mask = rng.uniform(-1, 1, B * H * W).astype(np.float32) if masked else None
    xz = x.astype(np.float64)
    if masked:
        xz = xz * (~(mask.reshape(B, H, W, 1) < -0.25))
    pad = k // 2
    xp = np.pad(xz, ((0, 0), (pad, pad), (pad, pad), (0, 0)))
    ref = np.zeros((k, k, Cin, Cout))
    for dy in range(k):
        for dx in range(k):
            ref[dy, dx] = np.einsum('bhwc,bhwd->cd', xp[:, dy:dy + H, dx:dx + W], g.astype(np.float64))
    err = {}
    for mode in (0, 2):
        try:
            assert be.dbg.dr_dbg_force_x3(mode) == 0
            dw = be.wgrad(x, g, k, T, nsplit, mask, -0.25)
        finally:
            be.dbg.dr_dbg_force_x3(be.x3_default)
        err[mode] = np.abs(dw - ref).max() / np.abs(ref).max()
    assert err[2] < 2e-5 and err[0] < 2e-5, err
    assert err[2] < 6 * err[0] + 2e-7, err


@pytest.mark.parametrize('case', [(2, 8, 8, 131, 70, 3, True), (1, 9, 7, 259, 300, 5, False), (2, 6, 6, 132, 128, 1, True), (1, 5, 5, 129, 33, 2, False)],
                         ids=lambda c: 'x'.join(str(int(v)) for v in c))
def test_wgrad_tail_split_of_a_few_extra_input_channels(be, case):
    """Cin = a multiple of 128 plus 1..4 channels (the heads' comb|uvd inputs: 515 = 512 + u, v, d; 131): the leading channels on
    conv_wgrad_x3_kernel<128>, the last ones by conv_wgrad_tail_kernel, both into the same slabs (conv_wgrad.h; T = 129 in the debug
    entry = the executor's split) -- against the fp64 einsum, masked input rows included, slabs that end inside a pixel step too."""
    B, H, W, Cin, Cout, nsplit, masked = case
    rng = np.random.default_rng(sum(case[:5]) + 5)
    x = (rng.standard_normal((B, H, W, Cin)) * np.exp(rng.uniform(-4, 4, (B, H, W, Cin)))).astype(np.float32)
    g = (rng.standard_normal((B, H, W, Cout)) * np.exp(rng.uniform(-4, 4, (B, H, W, Cout)))).astype(np.float32)
    mask = rng.uniform(-1, 1, B * H * W).astype(np.float32) if masked else None
    xz = x.astype(np.float64)
    if masked:
        xz = xz * (~(mask.reshape(B, H, W, 1) < -0.25))
    ref = np.einsum('bhwc,bhwd->cd', xz, g.astype(np.float64))[None, None]
    dw = be.wgrad(x, g, 1, 129, nsplit, mask, -0.25)
    assert dw.shape == (1, 1, Cin, Cout)
    assert np.abs(dw - ref).max() / np.abs(ref).max() < 2e-5
    # the tail rows on their own scale too (three channels among hundreds must not hide behind the tensor's maximum)
    t = Cin - Cin % 128
    assert np.abs(dw[0, 0, t:] - ref[0, 0, t:]).max() / np.abs(ref[0, 0, t:]).max() < 2e-5


def test_wgrad_seeded_shape_sweep(be):
    """A seeded sweep over small random weight-gradient problems -- odd image sides, ragged channel counts, both channel tiles and
    the kernel-row variant, any number of slabs, masked input rows -- against the fp64 einsum of the definition (index math: slab
    ends inside a pixel step, ragged last channel tiles, tap masks at the border, pad channels as poison)."""
    rng = np.random.default_rng(20240928)
    for case in range(60 if be.name == 'emu' else 80):
        B, H, W = int(rng.integers(1, 4)), int(rng.integers(1, 12)), int(rng.integers(1, 12))
        k = int(rng.choice([1, 3]))
        Cin = int(rng.choice([3, 16, 19, 33, 64, 70, 96, 131]))
        Cout = int(rng.choice([5, 14, 32, 42, 65, 78, 96, 128, 131]))
        Ts = [64, 128] + ([96] if (k == 3 and 64 < Cin <= 96 and 64 < Cout <= 96) else [])    # the kernel-row variant serves 65..96 channels
        Ts += [161, 162] if k == 3 else [163, 164, 165]                                        # the 16x16-tile kernels take any channel counts
        T = int(rng.choice(Ts))
        nsplit = int(rng.integers(1, 9))
        masked = bool(rng.random() < 0.3)
        x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
        g = rng.standard_normal((B, H, W, Cout)).astype(np.float32)
        mask = rng.uniform(-1, 1, B * H * W).astype(np.float32) if masked else None
        dw = be.wgrad(x, g, k, T, nsplit, mask, -0.25)
        xz = x.astype(np.float64)
        if masked:
            xz = xz * (~(mask.reshape(B, H, W, 1) < -0.25))
        pad = k // 2
        xp = np.pad(xz, ((0, 0), (pad, pad), (pad, pad), (0, 0)))
        ref = np.zeros((k, k, Cin, Cout))
        for dy in range(k):
            for dx in range(k):
                ref[dy, dx] = np.einsum('bhwc,bhwd->cd', xp[:, dy:dy + H, dx:dx + W], g.astype(np.float64))
        assert np.abs(dw - ref).max() / (np.abs(ref).max() + 1e-12) < 2e-5, (case, B, H, W, Cin, Cout, k, T, nsplit, masked)


@pytest.mark.parametrize('case', [c for c in WGRAD_CASES if c[6] in (64, 128)], ids=lambda c: 'x'.join(map(str, c[:8])))      # (the kernel-row and 16x16-tile kernels are fp32)
def test_wgrad_bf16_kernel_direct(be, case):
    """conv_wgrad_bf16_kernel (v_mfma_f32_32x32x16_bf16 over pixel-contiguous, register-transposed tiles): the fp64
    einsum of the bf16-rounded x and g -- what is left is fp32 summation order (2e-5).  Same cases as the fp32 kernel:
    ragged channels, masks, non power-of-two images, slabs that end inside a 32-pixel step."""
    from tests.common import bf16_round
    B, H, W, Cin, Cout, k, T, nsplit, masked = case
    rng = np.random.default_rng(sum(case[:6]) + 1)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    g = rng.standard_normal((B, H, W, Cout)).astype(np.float32)
    mask = rng.uniform(-1, 1, B * H * W).astype(np.float32) if masked else None
    try:
        assert be.dbg.dr_dbg_force_bf16(1) == 0
        dw = be.wgrad(x, g, k, T, nsplit, mask, -0.25)
    finally:
        be.dbg.dr_dbg_force_bf16(0)
    xz = bf16_round(x).astype(np.float64)
    if masked:
        xz = xz * (~(mask.reshape(B, H, W, 1) < -0.25))
    pad = k // 2
    xp = np.pad(xz, ((0, 0), (pad, pad), (pad, pad), (0, 0)))
    gz = bf16_round(g).astype(np.float64)
    ref = np.zeros((k, k, Cin, Cout))
    for dy in range(k):
        for dx in range(k):
            ref[dy, dx] = np.einsum('bhwc,bhwd->cd', xp[:, dy:dy + H, dx:dx + W], gz)
    assert np.abs(dw - ref).max() / np.abs(ref).max() < 2e-5


def test_adam_clip_accumulate_kernel(be):
    """Feed IDENTICAL gradients to the engine's fused clip+Adam and to the oracle (two steps)."""
    from oracle import net, train
    from oracle.graph import NetConfig, param_specs
    cfg = NetConfig(1, 8, 2)
    params = net.init_params(cfg, 3)
    h = be.handle(cfg, 1, training=True)
    h.load_params(params)
    h.call('dr_finalize_params', be.stream)
    addr, n = h.flat('grad')
    _, write = _flat_rw(be, addr, n)
    rng = np.random.default_rng(0)
    tr = [(nm, sh) for nm, sh, t in param_specs(cfg) if t]
    po = {nm: params[nm].copy() for nm, _ in tr}
    m = {nm: np.zeros(sh, np.float32) for nm, sh in tr}
    v = {nm: np.zeros(sh, np.float32) for nm, sh in tr}
    for step in (1, 2):
        acc = {nm: (rng.standard_normal(sh) * 2.0).astype(np.float32) for nm, sh in tr}     # |g|/5 often > 0.2: clip active
        write(np.concatenate([acc[nm].reshape(-1) for nm, _ in tr]))
        h.call('dr_apply_adam', C.c_float(1e-3), C.c_float(5.0), C.c_float(0.2), C.c_int64(step), be.stream)
        train.adam_step(po, m, v, acc, 1e-3, step, 5.0)
    got = h.read_params()
    for nm, _ in tr:
        np.testing.assert_allclose(got[nm], po[nm], rtol=2e-5, atol=2e-7, err_msg=nm)
    # packed weights follow the update: eval forward must now differ from the initial one only through them
    with pytest.raises(Exception):
        h.call('dr_apply_adam', C.c_float(1e-3), C.c_float(5.0), C.c_float(0.2), C.c_int64(0), be.stream)     # step < 1
    h.close()


def test_dropout_rng_mode_statistics(be):
    """DR_DROPOUT_RNG: about half of the 512-wide head activations are dropped, kept ones are doubled,
    a different seed gives a different mask, the same seed the same one."""
    from oracle.graph import conv_specs
    cfg, params, ndm, poses, cfgs, coms = _case(1, 64, 4, 1)
    h = be.handle(cfg, 1, training=True)
    h.load_params(params)
    h.call('dr_finalize_params', be.stream)
    d_dm = be.dev(ndm)
    name = [c.name for c in conv_specs(cfg) if c.cout == 512 and not c.bn][0]          # um_full 1

    def act(mode, seed):
        h.load_params(params)          # a train-mode forward advances the BatchReNorm state: reset it
        h.call('dr_finalize_params', be.stream)
        h.call('dr_forward_train', 1, be.ptr(d_dm), mode, None, C.c_uint64(seed), be.stream)
        return be.read_activation(h, name, (1, 32, 32, 512))
    a0, a1, a1b, a2 = act(0, 0), act(2, 1), act(2, 1), act(2, 2)
    np.testing.assert_array_equal(a1, a1b)
    pos = a0 > 0
    kept = (a1 != 0) & pos
    frac = kept.sum() / pos.sum()
    assert 0.47 < frac < 0.53, frac
    np.testing.assert_allclose(a1[kept], 2 * a0[kept], rtol=1e-6)
    assert (a1 != a2).mean() > 0.1
    h.close()


@pytest.mark.gpu
def test_training_trajectory_two_optimizer_steps(gpu):
    """End-to-end sequencing of the training loop (train_single_gpu.py:138-150): 2 optimizer steps x 2 micro-steps
    with BatchReNorm state carried between micro-steps, gradient accumulation, clip, Adam and weight re-packing.
    Adam turns near-zero gradient sign noise into +-lr parameter differences, so the check is on what the loop
    observes: the loss of every micro-step (engine vs oracle run side by side from the same start)."""
    from densereg_amd.parallel import DataParallelTrainer, learning_rate
    from oracle import net, pose, train
    from densereg_amd.data.synthetic import make_crops
    cfg, params, _, _, _, _ = _case(1, 64, 4, 3)
    J, B, SUB = cfg.num_jnt, 3, 2
    h = gpu.handle(cfg, B, training=True)
    h.load_params(params)
    h.call('dr_finalize_params', gpu.stream)
    h.call('dr_zero_grad', gpu.stream)
    p_o = {k: v.copy() for k, v in params.items()}
    names = [n for n in p_o if n.rsplit('/', 1)[1] in ('weights', 'biases', 'beta', 'gamma')]
    m = {n: np.zeros_like(p_o[n]) for n in names}
    v = {n: np.zeros_like(p_o[n]) for n in names}
    shadow, micro = {}, 0
    for step in range(2):
        acc = {n: np.zeros_like(p_o[n]) for n in names}
        for _ in range(SUB):
            dm, poses, cfgs, coms, _n = make_crops(B, 'icvl', seed=900 + micro)
            poses = np.ascontiguousarray(poses[:, :3 * J])
            ndm = pose.norm_dm(dm, coms)
            d_dm, d_pose, d_cfg, d_com, d_lo = gpu.dev(ndm), gpu.dev(poses), gpu.dev(cfgs), gpu.dev(coms), gpu.empty((4,))
            h.call('dr_forward_train', B, gpu.ptr(d_dm), 0, None, C.c_uint64(0), gpu.stream)
            h.call('dr_loss', B, gpu.ptr(d_dm), gpu.ptr(d_pose), gpu.ptr(d_cfg), gpu.ptr(d_com), gpu.ptr(d_lo), gpu.stream)
            h.call('dr_backward', B, gpu.stream)
            gpu.sync()
            lo, g, upd, _o = train.loss_and_grads(cfg, p_o, ndm, poses, cfgs, coms)
            np.testing.assert_allclose(gpu.host(d_lo), [lo[k] for k in ('hm', 'hm3', 'um', 'reg')], rtol=5e-3,
                                       err_msg='micro-step %d' % micro)
            net.bn_state_update(p_o, upd, zero_debias=True, shadow=shadow)
            for n in names:
                acc[n] += g[n]
            micro += 1
        lr = learning_rate(step, 'icvl', B, SUB)
        h.call('dr_apply_adam', C.c_float(lr), C.c_float(SUB), C.c_float(0.2), C.c_int64(step + 1), gpu.stream)
        h.call('dr_zero_grad', gpu.stream)
        train.adam_step(p_o, m, v, acc, lr, step + 1, float(SUB))
    got = h.read_params()
    frac_close = np.mean([np.mean(np.abs(got[n] - p_o[n]) < 1.5e-3) for n in names])
    assert frac_close > 0.97, frac_close                     # |delta| <= lr except where a tiny gradient flipped sign
    # and the engine can still run inference on the updated weights (eval fold is rebuilt lazily)
    hm, hm3, um = gpu.forward_eval(h, ndm)
    ep = net.forward_eval(cfg, p_o, ndm)
    assert np.isfinite(hm).all() and np.abs(hm - ep['hm_outs'][-1]).max() < 0.1
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize('B', [1, 7])
def test_ragged_batch_sizes(gpu, B):
    """B*H*W not a multiple of any tile: ragged last M tile in forward, dgrad and wgrad (max_batch > B)."""
    import torch
    from oracle import train
    cfg, params, ndm, poses, cfgs, coms = _case(1, 64, 4, B)
    h = gpu.handle(cfg, 8, training=True)
    h.load_params(params)
    h.call('dr_finalize_params', gpu.stream)
    d_dm, d_pose, d_cfg, d_com, d_lo = gpu.dev(ndm), gpu.dev(poses), gpu.dev(cfgs), gpu.dev(coms), gpu.empty((4,))
    h.call('dr_forward_train', B, gpu.ptr(d_dm), 0, None, C.c_uint64(0), gpu.stream)
    h.call('dr_loss', B, gpu.ptr(d_dm), gpu.ptr(d_pose), gpu.ptr(d_cfg), gpu.ptr(d_com), gpu.ptr(d_lo), gpu.stream)
    h.call('dr_zero_grad', gpu.stream)
    h.call('dr_backward', B, gpu.stream)
    gpu.sync()
    lo, _, _, _ = train.loss_and_grads(cfg, params, ndm, poses, cfgs, coms)
    np.testing.assert_allclose(gpu.host(d_lo), [lo[k] for k in ('hm', 'hm3', 'um', 'reg')], rtol=2e-4)
    g = flat_grads_by_name(gpu, h, cfg)
    _, g64, _, _ = train.loss_and_grads(cfg, params, ndm, poses, cfgs, coms, dtype=torch.float64)
    err = np.array([np.abs(g[n] - g64[n]).max() / (np.abs(g64[n]).max() + 1e-12) for n in g64])
    assert err.max() < 6e-2 and np.median(err) < 1e-2
    h.close()


def _engine_switches(be, h, cfg, B):
    """The discrete decisions of the ENGINE's last training forward, as the oracle takes them (oracle/net.py, TorchOps.switches):
    every ReLU's open units and every stored conv output (for the max-pools' argmax).  A BatchReNorm layer's ReLU is decided by
    raw * scale + shift > 0 -- the sign of the exact value, which is what the engine's fused multiply-add rounds -- from the layer's
    raw output and the multiply-add the forward applied (dr_read_activation "<scope>#raw" / "#fold"); a bias layer's by its stored
    output (no residual add is fused behind a bias layer's ReLU)."""
    from oracle.graph import conv_specs
    specs = {c.name: c for c in conv_specs(cfg)}

    def relu(name):
        c = specs[name]
        if c.bn:
            raw = be.read_activation(h, name + '#raw', (B, c.h_out, c.w_out, c.cout)).astype(np.float64)
            fold = be.read_activation(h, name + '#fold', (2 * c.cout,)).astype(np.float64)
            return raw * fold[:c.cout] + fold[c.cout:] > 0
        return be.read_activation(h, name, (B, c.h_out, c.w_out, c.cout)) > 0

    def act(name):
        c = specs[name]
        return be.read_activation(h, name, (B, c.h_out, c.w_out, c.cout))
    return {'relu': relu, 'act': act}


def _switch_free_gradient_check(be, cfg, params, ndm, poses, cfgs, coms, bar):
    """Arithmetic error alone: the engine's fp32 training micro-step against the oracle's fp64 autograd evaluated WITH THE ENGINE'S
    SWITCHES -- the same ReLU units open, the same max-pool winners (model/hourglass_um_crop_tiny.py:323-371 loss, um_v1.py graph).
    The bars of _run_step compare two evaluations that each take their own branches: an fp32 network flips a few of its ~10^7 ReLU
    / max-pool decisions against fp64, and one flip moves a gradient tensor by 1e-3 .. 1e-2 -- which is why those bars cannot be
    tighter than 6e-2.  With the switches injected nothing discrete is left, and a systematic 1 % error of any tensor (a wrong
    scale, a missing term, a mis-rounded operand split) would stand two orders of magnitude above the bar."""
    import torch
    from oracle import train
    B = ndm.shape[0]
    h = be.handle(cfg, B, training=True)
    h.load_params(params)
    h.call('dr_finalize_params', be.stream)
    d_dm, d_pose, d_cfg, d_com, d_lo = be.dev(ndm), be.dev(poses), be.dev(cfgs), be.dev(coms), be.empty((4,))
    h.call('dr_forward_train', B, be.ptr(d_dm), 0, None, C.c_uint64(0), be.stream)
    h.call('dr_loss', B, be.ptr(d_dm), be.ptr(d_pose), be.ptr(d_cfg), be.ptr(d_com), be.ptr(d_lo), be.stream)
    h.call('dr_zero_grad', be.stream)
    h.call('dr_backward', B, be.stream)
    be.sync()
    g = flat_grads_by_name(be, h, cfg)
    lo64, g64, _, _ = train.loss_and_grads(cfg, params, ndm, poses, cfgs, coms, dtype=torch.float64, switches=_engine_switches(be, h, cfg, B))
    np.testing.assert_allclose(be.host(d_lo), [lo64[k] for k in ('hm', 'hm3', 'um', 'reg')], rtol=2e-5)
    names, _, l2, cos = grad_metrics(g, g64)
    w = int(np.argmax(l2))
    print('switch-free gradient vs the fp64 oracle, per tensor rel-L2: max %.2e (%s) median %.2e | cosine min %.9f' % (l2.max(), names[w], np.median(l2), cos.min()))
    # for the record: the same fp64 oracle taking its OWN branches
    _, g64own, _, _ = train.loss_and_grads(cfg, params, ndm, poses, cfgs, coms, dtype=torch.float64)
    _, _, l2o, _ = grad_metrics(g, g64own)
    print('   ... against the fp64 oracle with its own switches: max %.2e median %.2e' % (l2o.max(), np.median(l2o)))
    assert l2.max() < bar, (l2.max(), names[w])
    h.close()


def test_gradient_without_switches_single_stack(be):
    """Emulator and GPU: S=1 F=64 J=4, one crop (three on the GPU).  The bar is 1e-4 per tensor (fp32 arithmetic of this depth
    measures 1e-6 .. 3e-5)."""
    cfg, params, ndm, poses, cfgs, coms = _case(1, 64, 4, 1 if be.name == 'emu' else 3)
    _switch_free_gradient_check(be, cfg, params, ndm, poses, cfgs, coms, 1e-4)


@pytest.mark.gpu
def test_gradient_without_switches_config3_b4(gpu):
    """BASELINE config 3's network (NYU S=2 F=128 J=14) at B=4, the kernels the headline times (x3 family on the big layers): every
    gradient tensor within 1e-4 relative L2 of the fp64 oracle that takes the engine's branches."""
    cfg, params, ndm, poses, cfgs, coms = _case(2, 128, 14, 4, 'nyu')
    _switch_free_gradient_check(gpu, cfg, params, ndm, poses, cfgs, coms, 1e-4)
