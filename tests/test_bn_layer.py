"""Kernel-level pin of the train-mode BatchReNorm path: ONE conv -> BatchReNorm(train) -> ReLU (+ residual) layer through
``dr_dbg_bn_layer`` -- the executors' own launch logic (conv epilogue statistics, finalize or fused fold, apply; backward
reduce or the consumer-dgrad-fused sums, finalize, apply) -- against an fp64 autograd of the same layer
(``network/slim/ops.py:130-171``: biased moments, eps inside the sqrt, r / d clipped and stop-gradient, zero-debiased
moving averages).  Tolerance 1e-5 of each tensor's max (the whole-network tests accept 6e-2: a 1 % error in
``mean(g * yhat)`` would pass there and fails here).  ``[emu]`` on CPU fibers (small shapes), ``[gpu]`` on an MI355X.
"""
import ctypes as C

import numpy as np
import pytest

from densereg_amd import _lib

BACKENDS = [pytest.param('emu'), pytest.param('gpu', marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def be(request):
    return request.getfixturevalue(request.param)


def _reference(x, w, gamma, beta, mm, mv, r_max, d_max, relu, res, dout, gr, wr):
    """fp64 torch autograd of conv -> BatchReNorm(train) -> relu (+res) [-> consumer conv]."""
    import torch
    import torch.nn.functional as F
    t = lambda a: None if a is None else torch.from_numpy(np.asarray(a, np.float64))
    xt, wt = t(x), t(w)
    k = w.shape[0]
    raw = F.conv2d(xt.permute(0, 3, 1, 2), wt.permute(3, 2, 0, 1), padding=k // 2).permute(0, 2, 3, 1)
    raw = raw.detach().requires_grad_(True)
    g_, b_ = t(gamma).requires_grad_(True), t(beta).requires_grad_(True)
    eps = 1e-3
    mean = raw.mean((0, 1, 2))
    var = ((raw - mean) ** 2).mean((0, 1, 2))                     # biased (tf.nn.moments)
    std = torch.sqrt(var + eps)
    mstd = torch.sqrt(t(mv) + eps)
    r = torch.clamp(std / mstd, 1.0 / r_max, r_max).detach()      # stop_gradient (ops.py:139-140)
    d = torch.clamp((mean - t(mm)) / mstd, -d_max, d_max).detach()
    yhat = (raw - mean) / std
    out = (yhat * r + d) * g_ + b_
    if relu:
        out = torch.relu(out)
    rt_ = None
    if res is not None:
        rt_ = t(res).requires_grad_(True)
        out = out + rt_
    out.retain_grad()
    loss = 0.0
    if dout is not None:
        loss = loss + (out * t(dout)).sum()
    if gr is not None:
        kr = wr.shape[0]
        z = F.conv2d(out.permute(0, 3, 1, 2), t(wr).permute(3, 2, 0, 1), padding=kr // 2).permute(0, 2, 3, 1)
        loss = loss + (z * t(gr)).sum()
    loss.backward()
    om = 1.0 - 0.99                                                # zero-debiased first update: biased = value*(1-decay),
    corr = 1.0 - 0.99 ** 1                                         # divided by 1 - decay^1 -> the batch value itself
    mm_next = (0.0 - (0.0 - mean.detach()) * om) / corr
    mv_next = (0.0 - (0.0 - var.detach()) * om) / corr
    n = lambda v: None if v is None else v.detach().numpy()
    return dict(raw=n(raw), y=n(out), mean=n(mean), istd=n(1.0 / std), r=n(r), d=n(d), mm_next=n(mm_next), mv_next=n(mv_next),
                dout=n(out.grad), draw=n(raw.grad), dgamma=n(g_.grad), dbeta=n(b_.grad), dres=None if rt_ is None else n(rt_.grad))


def _run(be, B, H, W, Cin, Cout, k, relu=True, with_res=False, consumer=None, seed=0, r_max=3.0, d_max=5.0, both=False,
         return_raw_draw=False):
    rng = np.random.default_rng(seed)
    cs, x_cs = -(-Cout // 4) * 4, -(-Cin // 4) * 4
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    gamma = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    beta = (0.3 * rng.standard_normal(Cout)).astype(np.float32)
    # moving stats placed so that some channels clip r / d and others do not
    mm = (0.4 * rng.standard_normal(Cout)).astype(np.float32)
    mv = rng.uniform(0.05, 4.0, Cout).astype(np.float32)
    res = rng.standard_normal((B, H, W, Cout)).astype(np.float32) if with_res else None
    dout = gr = wr = None
    if consumer is None or both:
        dout = rng.standard_normal((B, H, W, Cout)).astype(np.float32)
    if consumer is not None:
        kr, Cr = consumer
        wr = (rng.standard_normal((kr, kr, Cout, Cr)) / np.sqrt(kr * kr * Cout)).astype(np.float32)
        gr = rng.standard_normal((B, H, W, Cr)).astype(np.float32)

    def padded(a, stride, fill=np.nan):
        if a is None:
            return None
        out = np.full(a.shape[:-1] + (stride,), fill, np.float32)      # pad channels are poison: must not be read
        out[..., :a.shape[-1]] = a
        return out
    d = {n: be.dev(v) for n, v in dict(x=padded(x, x_cs), w=w, gamma=gamma, beta=beta, mm=mm, mv=mv).items()}
    d_res = None if res is None else be.dev(padded(res, cs))
    d_dout = None if dout is None else be.dev(padded(dout, cs))
    gr_cs = 0
    d_gr = d_wr = None
    if consumer is not None:
        gr_cs = -(-consumer[1] // 4) * 4
        d_gr, d_wr = be.dev(padded(gr, gr_cs)), be.dev(wr)
    M = B * H * W
    o = {n: be.dev(np.full((M, cs), -777.0, np.float32)) for n in ('y', 'raw', 'dout_used', 'draw', 'dres')}
    ov = {n: be.dev(np.zeros(Cout, np.float32)) for n in ('mm_next', 'mv_next', 'dgamma', 'dbeta')}
    bnc = be.dev(np.zeros((4, Cout), np.float32))
    a = _lib.DbgBnArgs(B, H, W, Cin, Cout, k, be.ptr(d['x']), x_cs, be.ptr(d['w']), be.ptr(d['gamma']), be.ptr(d['beta']),
                       be.ptr(d['mm']), be.ptr(d['mv']), r_max, d_max, int(relu), be.ptr(d_res), be.ptr(d_dout),
                       be.ptr(d_gr), gr_cs, be.ptr(d_wr), 0 if consumer is None else consumer[0], 0 if consumer is None else consumer[1],
                       be.ptr(o['y']), be.ptr(o['raw']), be.ptr(bnc), be.ptr(ov['mm_next']), be.ptr(ov['mv_next']),
                       be.ptr(o['dout_used']), be.ptr(o['draw']), be.ptr(ov['dgamma']), be.ptr(ov['dbeta']),
                       be.ptr(o['dres']) if with_res else None, 0, 0)
    rc = be.dbg.dr_dbg_bn_layer(C.byref(a), be.stream)
    assert rc == 0, rc
    be.sync()
    if return_raw_draw:                       # the draw / y buffers as the kernels left them (bf16-storage check of the training tests)
        return be.host(o['draw']).reshape(M, cs).copy(), be.host(o['y']).reshape(M, cs).copy()
    ref = _reference(x, w, gamma, beta, mm, mv, r_max, d_max, relu, res, dout, gr, wr)
    got = {n: be.host(v).reshape(B, H, W, cs)[..., :Cout] for n, v in o.items()}
    gv = {n: be.host(v) for n, v in ov.items()}
    g_bnc = be.host(bnc)

    def close(name, a_, b_, tol=1e-5):
        sc = np.abs(b_).max() + 1e-12
        err = np.abs(a_ - b_).max() / sc
        assert err <= tol, '%s: %.3e of the tensor max (rows fwd %d bwd %d)' % (name, err, a.fwd_rows, a.bwd_rows)
    close('raw', got['raw'], ref['raw'])
    close('y', got['y'], ref['y'])
    for i, n in enumerate(('mean', 'istd', 'r', 'd')):
        close('bnc.' + n, g_bnc[i], ref[n], 2e-6 if n != 'mean' else 1e-5)
    close('mm_next', gv['mm_next'], ref['mm_next'])
    close('mv_next', gv['mv_next'], ref['mv_next'])
    close('dout', got['dout_used'], ref['dout'])
    close('draw', got['draw'], ref['draw'])
    close('dgamma', gv['dgamma'], ref['dgamma'])
    close('dbeta', gv['dbeta'], ref['dbeta'])
    if with_res:
        close('dres', got['dres'], ref['dres'])
    clipped = (np.abs(ref['r'] - 3.0) < 1e-12) | (np.abs(ref['r'] - 1 / 3.0) < 1e-12) | (np.abs(np.abs(ref['d']) - d_max) < 1e-12)
    return a.fwd_rows, a.bwd_rows, clipped


# (B, H, W, Cin, Cout, k): ragged channel counts of the hm3 / um heads (65, 78, 131), a few rows and many
CASES_SMALL = [(1, 8, 8, 19, 65, 1), (2, 4, 4, 40, 78, 3), (3, 8, 8, 16, 131, 1)]
CASES_GPU = [(4, 32, 32, 131, 65, 1), (4, 32, 32, 78, 78, 3), (6, 32, 32, 160, 131, 1), (40, 8, 8, 64, 64, 3), (5, 16, 16, 128, 64, 1),
             (40, 32, 32, 128, 256, 1)]


def _cases(be):
    return CASES_SMALL if be.name == 'emu' else CASES_SMALL + CASES_GPU


def test_bn_layer_forward_backward_own_reduce(be):
    """dOut given: the layer runs bn_bwd_reduce + finalize (or the fused fold on few rows) + apply; with and without the
    residual branch (dres written inside the apply pass)."""
    paths = set()
    for i, (B, H, W, Cin, Cout, k) in enumerate(_cases(be)):
        fr, br, clipped = _run(be, B, H, W, Cin, Cout, k, relu=True, with_res=(i % 2 == 0), seed=i)
        paths.add((fr <= 48, br <= 48))
        assert 0 < clipped.sum() < clipped.size            # both the clipped and the unclipped renorm branch are exercised
    if be.name == 'gpu':
        assert paths >= {(True, True), (False, False)}, paths          # fused fold and finalize launch both ran


def test_bn_layer_finalize_launch_with_several_waves_per_channel(be):
    """The finalize launches (one workgroup per channel; a wave per 512 partial rows, train_kernels.h: bn_finalize_split) with
    the threshold lowered to 16 rows: the 50 statistics rows of a 2x56x57 layer fold through four waves, forward and backward
    (the backward sums from a consumer's dgrad; the layer's own reduce pass writes few rows: the fused fold).  At the default
    threshold the 40x32x32x256 layer of CASES_GPU (640 rows: two waves) runs that path in the tests around this one -- a seed
    there is a seed without a ReLU knife edge: 2 of that layer's 10 485 760 pre-activations within 4e-8 of zero flip the mask
    against fp64 with seed 5, and one flipped element is 1.7e-2 of draw's max."""
    assert be.dbg.dr_dbg_bn_finalize_rows(16) == 0
    try:
        fr, br, clipped = _run(be, 2, 56, 57, 8, 12, 1, relu=True, with_res=True, seed=3)
        assert fr > 48, fr                                     # (48 rows and fewer: no finalize launch at all)
        fr, br, _ = _run(be, 2, 56, 57, 8, 12, 1, relu=True, consumer=(1, 9), seed=4)
        assert fr > 48 and br > 48, (fr, br)
    finally:
        be.dbg.dr_dbg_bn_finalize_rows(0)


def test_bn_layer_backward_sums_from_the_consumers_dgrad(be):
    """A consumer conv given: its dgrad launch writes dOut AND the layer's sum(g), sum(g*yhat) partial rows in the epilogue
    (plan_backward's single-reader path) -- against the same fp64 autograd, not against the unfused path."""
    consumers = [(1, 33), (3, 20), (1, 64)]
    for i, (B, H, W, Cin, Cout, k) in enumerate(_cases(be)):
        _run(be, B, H, W, Cin, Cout, k, relu=True, consumer=consumers[i % 3], seed=100 + i)
        # ... and as the LAST writer: dOut already holds another reader's gradient, the dgrad accumulates onto it
        _run(be, B, H, W, Cin, Cout, k, relu=True, with_res=(i % 2 == 1), consumer=consumers[(i + 1) % 3], seed=200 + i, both=True)


def test_bn_layer_without_relu_and_pure_batchnorm_schedule(be):
    """r_max = 1, d_max = 0 (the schedule's start, ops.py:141-152): plain batch normalisation; no ReLU."""
    B, H, W, Cin, Cout, k = CASES_SMALL[0]
    _, _, clipped = _run(be, B, H, W, Cin, Cout, k, relu=False, r_max=1.0, d_max=0.0, seed=7)
    assert clipped.all()


def test_bias_conv_backward_from_the_readers_dgrad(be):
    """um_full 1 / 2 (bias + ReLU + dropout, um_v1.py:155-165): the single reader's dgrad writes the gradient wrt the conv's
    pre-activation, dOut * 2 * [out > 0], and the bias column sums -- against an fp64 autograd of relu-dropout -> conv."""
    import torch
    import torch.nn.functional as F
    cases = [(1, 8, 8, 70, 33, 1, 2.0), (2, 4, 4, 37, 48, 3, 1.0)]
    if be.name == 'gpu':
        cases += [(4, 32, 32, 512, 48, 1, 2.0), (3, 32, 32, 512, 512, 1, 2.0)]
    for i, (B, H, W, Cc, Cr, kr, factor) in enumerate(cases):
        rng = np.random.default_rng(50 + i)
        cs, gr_cs = -(-Cc // 4) * 4, -(-Cr // 4) * 4
        pre = rng.standard_normal((B, H, W, Cc))
        keep = rng.integers(0, 2, pre.shape) if factor == 2.0 else np.ones(pre.shape)
        out = (np.maximum(pre, 0) * keep * factor).astype(np.float32)            # what the forward stored
        wr = (rng.standard_normal((kr, kr, Cc, Cr)) / np.sqrt(kr * kr * Cc)).astype(np.float32)
        gr = rng.standard_normal((B, H, W, Cr)).astype(np.float32)
        ot = torch.from_numpy(out.astype(np.float64)).requires_grad_(True)
        z = F.conv2d(ot.permute(0, 3, 1, 2), torch.from_numpy(wr.astype(np.float64)).permute(3, 2, 0, 1), padding=kr // 2).permute(0, 2, 3, 1)
        (z * torch.from_numpy(gr.astype(np.float64))).sum().backward()
        g_ref = ot.grad.numpy() * factor * (out > 0)
        pad = lambda a, stride: np.concatenate([a, np.full(a.shape[:-1] + (stride - a.shape[-1],), np.nan, np.float32)], -1)
        d_out, d_gr, d_wr = be.dev(pad(out, cs)), be.dev(pad(gr, gr_cs)), be.dev(wr)
        d_g, d_b = be.dev(np.full((B * H * W, cs), -777.0, np.float32)), be.dev(np.full(Cc, 0.5, np.float32))
        rc = be.dbg.dr_dbg_act_dgrad(B, H, W, Cc, Cr, kr, be.ptr(d_out), be.ptr(d_gr), gr_cs, be.ptr(d_wr), factor, be.ptr(d_g),
                                     be.ptr(d_b), be.stream)
        assert rc == 0, rc
        be.sync()
        g = be.host(d_g).reshape(B, H, W, cs)[..., :Cc]
        assert np.abs(g - g_ref).max() / np.abs(g_ref).max() < 1e-5
        np.testing.assert_allclose(be.host(d_b) - 0.5, g_ref.sum((0, 1, 2)), rtol=1e-5, atol=1e-5 * np.abs(g_ref).sum((0, 1, 2)).max())


def test_bn_layer_seeded_sweep(be):
    """Seeded random layers through dr_dbg_bn_layer against the fp64 autograd (1e-5 of each tensor's max, as above): odd image
    sides, ragged channel counts, 1x1 and 3x3, with / without ReLU and residual, own reduce pass or sums from a consumer's dgrad
    (single reader or last writer), row counts on both sides of the fused-fold threshold."""
    rng = np.random.default_rng(777)
    n = 10 if be.name == 'emu' else 24
    for i in range(n):
        big = be.name == 'gpu' or i % 5 == 0                          # some layers beyond 48 partial rows (finalize launch)
        B = int(rng.integers(1, 5))
        H, W = (int(rng.integers(20, 40)), int(rng.integers(20, 40))) if big else (int(rng.integers(1, 10)), int(rng.integers(1, 10)))
        Cin = int(rng.choice([3, 16, 19, 40, 64]))
        Cout = int(rng.choice([5, 14, 33, 65, 78, 128, 131]))
        k = int(rng.choice([1, 3]))
        relu = bool(rng.random() < 0.7)
        with_res = bool(rng.random() < 0.3)
        mode = int(rng.integers(0, 3))                                # 0 own reduce, 1 consumer dgrad, 2 consumer as the last writer
        consumer = None if mode == 0 else (int(rng.choice([1, 3])), int(rng.choice([8, 33, 64])))
        _run(be, B, H, W, Cin, Cout, k, relu=relu, with_res=with_res, consumer=consumer, both=(mode == 2), seed=1000 + i)


def test_bn_layer_lookback_handoff_opt_in(be, monkeypatch):
    """DR_BN_LOOKBACK=1 (opt-in; measured slower on MI355X, kept correct): the apply launches carry producer workgroups that
    fold the partial rows and hand the coefficients to the streaming workgroups behind a counter, no finalize launch."""
    monkeypatch.setenv('DR_BN_LOOKBACK', '1')
    for i, case in enumerate([(8, 32, 32, 8, 65, 1), (8, 32, 32, 16, 131, 1)] if be.name == 'emu' else [(8, 32, 32, 8, 65, 1), (40, 32, 32, 128, 256, 1)]):
        fr, br, _ = _run(be, *case, relu=True, with_res=(i == 0), seed=300 + i)
        assert fr > 48 and br > 48            # the many-rows path, where the hand-off replaces the finalize launch
