"""Micro-batch groups (``dr_set_groups``): the G micro-steps of one accumulation window as ONE pass of launches.  Same arithmetic
per micro-batch as the reference's loop (train_single_gpu.py:138-150) -- statistics, r / d, the clip schedule, the moving-statistics
chain, the gradient sum -- with sums taken over other tile shapes.  Checked against the oracle's chained micro-steps (emulator and
GPU) and, on the GPU, against the engine's own micro-step loop."""
import ctypes as C

import numpy as np
import pytest

from tests.common import _flat_rw

BACKENDS = [pytest.param('emu'), pytest.param('gpu', marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def be(request):
    return request.getfixturevalue(request.param)


def _case(be, G, Bg):
    from densereg_amd.data.synthetic import make_crops
    from oracle import net, pose
    from oracle.graph import NetConfig
    S, F, J = (1, 8, 2) if be.name == 'emu' else (2, 64, 5)
    cfg = NetConfig(S, F, J)
    dm, poses, cfgs, coms, _ = make_crops(G * Bg, 'icvl', seed=77)
    ndm = pose.norm_dm(dm, coms)
    params = net.make_test_params(cfg, ndm[:2], seed=5)
    return cfg, params, (ndm, np.ascontiguousarray(poses[:, :3 * J]), cfgs, coms)


def _run(be, cfg, params, data, G, Bg, fused, windows=1, bf16=False):
    """`windows` accumulation windows of G micro-batches (the same data each time: the state chain is what differs); returns
    (losses [windows*G][4], accumulated gradient of the last window, parameters incl. BatchReNorm state)."""
    B = G * Bg
    h = be.handle(cfg, B, training=True)
    if bf16:
        h.call('dr_set_precision', 1)
    h.load_params(params)
    h.call('dr_finalize_params', be.stream)
    bufs = [be.dev(np.ascontiguousarray(a)) for a in data]
    losses = []
    for w in range(windows):
        h.call('dr_zero_grad', be.stream)
        if fused:
            h.call('dr_set_groups', G)
            d_lo = be.empty((G, 4))
            h.call('dr_forward_train', B, be.ptr(bufs[0]), 0, None, C.c_uint64(9), be.stream)
            h.call('dr_loss', B, be.ptr(bufs[0]), be.ptr(bufs[1]), be.ptr(bufs[2]), be.ptr(bufs[3]), be.ptr(d_lo), be.stream)
            h.call('dr_backward', B, be.stream)
            be.sync()
            losses.extend(be.host(d_lo).reshape(G, 4).copy())
        else:
            for g in range(G):
                sl = [be.dev(np.ascontiguousarray(a[g * Bg:(g + 1) * Bg])) for a in data]
                d_lo = be.empty((4,))
                h.call('dr_forward_train', Bg, be.ptr(sl[0]), 0, None, C.c_uint64(9), be.stream)
                h.call('dr_loss', Bg, be.ptr(sl[0]), be.ptr(sl[1]), be.ptr(sl[2]), be.ptr(sl[3]), be.ptr(d_lo), be.stream)
                h.call('dr_backward', Bg, be.stream)
                be.sync()
                losses.append(be.host(d_lo).copy())
    addr, n = h.flat('grad')
    grad = _flat_rw(be, addr, n)[0]().copy()
    out = np.array(losses), grad, h.read_params()
    h.close()
    return out


def _compare(lo_s, g_s, p_s, lo_f, g_f, p_f, params):
    assert np.isfinite(lo_f).all() and np.isfinite(g_f).all()
    np.testing.assert_allclose(lo_f, lo_s, rtol=2e-5)
    for k in p_s:
        if 'moving' in k:
            np.testing.assert_allclose(p_f[k], p_s[k], rtol=2e-5, atol=1e-6 * max(1.0, float(np.abs(p_s[k]).max())), err_msg=k)
        elif k.endswith(('r_max', 'd_max', 'curr_t')):
            np.testing.assert_array_equal(p_f[k], p_s[k], err_msg=k)
    # the moving statistics moved (the chain is really G updates long)
    moved = [k for k in p_s if 'moving_mean' in k and np.abs(p_s[k] - params[k]).max() > 0]
    assert len(moved) > 0
    scale = float(np.abs(g_s).max())
    err = np.abs(g_f - g_s)
    assert err.max() <= 2e-3 * scale, (err.max(), scale)
    assert np.median(err) <= 1e-5 * scale


@pytest.mark.gpu
def test_groups_match_sequential_micro_steps(gpu):
    """(GPU only: on the emulator the window pass is checked against the oracle below -- the same statement at half the fibers)"""
    be = gpu
    G, Bg = 3, 8
    cfg, params, data = _case(be, G, Bg)
    windows = 2
    lo_s, g_s, p_s = _run(be, cfg, params, data, G, Bg, fused=False, windows=windows)
    lo_f, g_f, p_f = _run(be, cfg, params, data, G, Bg, fused=True, windows=windows)
    assert lo_f.shape == (windows * G, 4)
    # the micro-batches differ, and so do their losses: the rows are really per group
    assert np.abs(lo_f[0, :3] - lo_f[1, :3]).max() > 0
    _compare(lo_s, g_s, p_s, lo_f, g_f, p_f, params)


@pytest.mark.gpu
def test_groups_match_sequential_micro_steps_on_the_bf16_matrix_cores(gpu):
    """The same comparison with ``dr_set_precision(bf16)``.  Operands are rounded to bf16 element by element whatever the tile, but
    the fp32 sums behind them are taken in another order, and a last-bit difference in an activation flips bf16 roundings (2^-8
    relative) in the next layer's operands: window pass and micro-step loop agree to bf16 noise, not to fp32 rounding -- the bars are
    those of two bf16 evaluations of the same graph (measured on MI355X: losses 2e-3, moving statistics 9e-3 of their max, gradient
    3.2e-2 of its max on the worst element, median 3.6e-5; `profiles/r03_groups_test_gpu.log`).  What the test pins is the plumbing: the bf16 storage decisions (activations,
    dRaw) have to agree with the tiles the grouped launches get (conv_tile_id with grp_rows), every launch has to succeed."""
    be = gpu
    G, Bg = 3, 8
    cfg, params, data = _case(be, G, Bg)
    lo_s, g_s, p_s = _run(be, cfg, params, data, G, Bg, fused=False, bf16=True)
    lo_f, g_f, p_f = _run(be, cfg, params, data, G, Bg, fused=True, bf16=True)
    assert np.isfinite(lo_f).all() and np.isfinite(g_f).all()
    scale = float(np.abs(g_s).max())
    err = np.abs(g_f - g_s)
    worst_state = max(float(np.abs(p_f[k] - p_s[k]).max() / max(1e-3, np.abs(p_s[k]).max())) for k in p_s if 'moving' in k)
    print('bf16 window vs bf16 micro-step loop: losses max rel %.2e, moving statistics %.2e of their max, gradient max err %.2e of max, median %.2e'
          % (np.abs(lo_f / lo_s - 1).max(), worst_state, err.max() / scale, np.median(err) / scale))
    np.testing.assert_allclose(lo_f, lo_s, rtol=1e-2)
    assert worst_state < 5e-2
    assert err.max() <= 1e-1 * scale and np.median(err) <= 5e-3 * scale
    for k in p_s:
        if k.endswith(('r_max', 'd_max', 'curr_t')):
            np.testing.assert_array_equal(p_f[k], p_s[k], err_msg=k)


@pytest.mark.gpu
def test_groups_window_is_bit_reproducible_at_eight_groups_with_dropout(gpu):
    """The largest window (G = 8) with ``DR_DROPOUT_RNG``: finite, every micro-batch has its own loss row, the same bits on two
    fresh handles (no floating-point atomics anywhere on the path, per-group sums in fixed order), another seed gives other masks,
    and a repeated loss + backward on the same forward doubles the accumulated gradient."""
    from densereg_amd.data.synthetic import make_crops
    from oracle import net, pose
    from oracle.graph import NetConfig
    be = gpu
    G, Bg, S, F, J = 8, 8, 1, 64, 14
    B = G * Bg
    cfg = NetConfig(S, F, J)
    dm, poses, cfgs, coms, _ = make_crops(B, 'nyu', seed=17)
    data = (pose.norm_dm(dm, coms), np.ascontiguousarray(poses[:, :3 * J]), cfgs, coms)
    params = net.make_test_params(cfg, data[0][:4], seed=9)

    def run(seed, twice=False):
        h = be.handle(cfg, B, training=True)
        h.load_params(params)
        h.call('dr_finalize_params', be.stream)
        h.call('dr_zero_grad', be.stream)
        h.call('dr_set_groups', G)
        d = [be.dev(np.ascontiguousarray(a)) for a in data]
        d_lo = be.empty((G, 4))
        h.call('dr_forward_train', B, be.ptr(d[0]), 2, None, C.c_uint64(seed), be.stream)
        for _ in range(2 if twice else 1):
            h.call('dr_loss', B, be.ptr(d[0]), be.ptr(d[1]), be.ptr(d[2]), be.ptr(d[3]), be.ptr(d_lo), be.stream)
            h.call('dr_backward', B, be.stream)
        be.sync()
        addr, n = h.flat('grad')
        out = be.host(d_lo).reshape(G, 4).copy(), _flat_rw(be, addr, n)[0]().copy(), h.read_params()
        h.close()
        return out
    lo_a, g_a, p_a = run(5)
    lo_b, g_b, p_b = run(5)
    assert np.isfinite(lo_a).all() and np.isfinite(g_a).all() and np.abs(g_a).max() > 0
    assert len({tuple(r[:3]) for r in lo_a.tolist()}) == G              # eight different micro-batches, eight different rows
    assert np.all(lo_a[:, 3] == lo_a[0, 3])                             # the regulariser is the same for all of them
    np.testing.assert_array_equal(lo_a, lo_b)
    np.testing.assert_array_equal(g_a, g_b)
    for k in p_a:
        np.testing.assert_array_equal(p_a[k], p_b[k], err_msg=k)
    lo_c, g_c, _ = run(6)
    assert np.abs(lo_c - lo_a).max() > 0                                # other seed, other masks
    _, g_2, _ = run(5, twice=True)
    np.testing.assert_allclose(g_2, 2.0 * g_a, rtol=0, atol=1e-4 * float(np.abs(g_a).max()))


def test_groups_reject_what_the_tiles_cannot_cut(be):
    from oracle.graph import NetConfig
    cfg = NetConfig(1, 8, 2)
    h = be.handle(cfg, 6, training=True)
    from densereg_amd._lib import DenseRegError
    with pytest.raises(DenseRegError):
        h.call('dr_set_groups', 9)
    h.call('dr_set_groups', 2)
    dm = be.dev(np.zeros((6, 128, 128), np.float32))
    with pytest.raises(DenseRegError):          # 5 crops are not two equal groups
        h.call('dr_forward_train', 5, be.ptr(dm), 0, None, C.c_uint64(0), be.stream)
    h.close()


def test_groups_window_against_the_oracles_chained_micro_steps(be):
    """The window pass against the ORACLE: G micro-steps of ``oracle.train.loss_and_grads`` with the BatchReNorm state update
    (``oracle.net.bn_state_update``: moving statistics with zero-debias, r_max / d_max / curr_t) between them, each with its own
    injected dropout masks -- losses per micro-batch, the summed gradient, and the state after the window."""
    from densereg_amd.data.synthetic import make_crops
    from oracle import net, pose, train
    from oracle.graph import NetConfig
    from tests.common import flat_grads_by_name
    G, Bg, S, F, J = (2, 8, 1, 8, 2) if be.name == 'emu' else (3, 8, 2, 32, 4)
    B = G * Bg
    cfg = NetConfig(S, F, J)
    dm, poses, cfgs, coms, _ = make_crops(B, 'nyu', seed=91)
    poses = np.ascontiguousarray(poses[:, :3 * J])
    ndm = pose.norm_dm(dm, coms)
    params = net.make_test_params(cfg, ndm[:4], seed=3)
    rng = np.random.default_rng(1)
    masks = [rng.integers(0, 2, (B, 32, 32, 512)).astype(np.uint8) for _ in range(2 * S)]      # per dropout layer, all B crops
    # oracle: the reference's loop
    p = {k: v.copy() for k, v in params.items()}
    shadow, want_lo, gsum = {}, [], None
    for g in range(G):
        sl = slice(g * Bg, (g + 1) * Bg)
        lo, gr, upd, _ = train.loss_and_grads(cfg, p, ndm[sl], poses[sl], cfgs[sl], coms[sl], dropout_masks=[m[sl] for m in masks])
        want_lo.append([lo[k] for k in ('hm', 'hm3', 'um', 'reg')])
        gsum = gr if gsum is None else {k: gsum[k] + gr[k] for k in gr}
        net.bn_state_update(p, upd, zero_debias=True, shadow=shadow)
    # engine: one pass
    h = be.handle(cfg, B, training=True)
    h.load_params(params)
    h.call('dr_finalize_params', be.stream)
    h.call('dr_zero_grad', be.stream)
    h.call('dr_set_groups', G)
    d = [be.dev(a) for a in (ndm, poses, cfgs, coms)]
    d_mask, d_lo = be.dev(np.ascontiguousarray(np.stack(masks))), be.empty((G, 4))
    if be.name == 'emu':
        # the BatchReNorm finalize launches give a micro-batch group several waves from 512 statistics rows on (train_kernels.h:
        # bn_finalize_split; the full-size window tests of tests/test_bench_shapes.py run there): from 16 rows on here, so the
        # larger layers of this small case take that path too (the emulator library carries the hook; process-global)
        assert be.dbg.dr_dbg_bn_finalize_rows(16) == 0
    try:
        h.call('dr_forward_train', B, be.ptr(d[0]), 1, be.ptr(d_mask), C.c_uint64(0), be.stream)
        h.call('dr_loss', B, be.ptr(d[0]), be.ptr(d[1]), be.ptr(d[2]), be.ptr(d[3]), be.ptr(d_lo), be.stream)
        h.call('dr_backward', B, be.stream)
        be.sync()
    finally:
        if be.name == 'emu':
            be.dbg.dr_dbg_bn_finalize_rows(0)
    np.testing.assert_allclose(be.host(d_lo).reshape(G, 4), np.array(want_lo), rtol=3e-4)
    got = h.read_params()
    for k in p:
        if 'moving' in k or k.endswith(('r_max', 'd_max', 'curr_t')):
            np.testing.assert_allclose(got[k], p[k], rtol=3e-4, atol=5e-5 * max(1.0, float(np.abs(p[k]).max())), err_msg=k)
    grads = flat_grads_by_name(be, h, cfg)
    e = np.array([np.abs(grads[n] - gsum[n]).max() / (np.abs(gsum[n]).max() + 1e-12) for n in gsum])
    print('window gradient vs the oracle (fp32 autograd, %d micro-steps summed): max %.2e median %.2e' % (G, e.max(), np.median(e)))
    assert e.max() < 1.6e-1 and np.median(e) < 2e-2, (e.max(), np.median(e))         # the bar of tests/test_train_parity.py (1)
    h.close()
