"""GPU-only parity at BASELINE.json's sizes (run with ``-m gpu`` on an MI355X).

* the four conv shapes that carry 59 % of the FLOPs, against an fp64 reference;
* config 2 (ICVL S=2 F=128 B=40): every head map and the voted xyz against the CPU oracle on the same
  seeded inputs -- BASELINE.json bar: mean-joint-error delta <= 0.1 mm;
* size-independent properties at full batch: batch-composition invariance (sample i of a B=40 batch
  == the same sample in a shuffled B=39 batch, bit-exact: eval mode has no cross-sample term and every output
  element is one fixed-order fma chain; a B=3 batch selects other tiles and agrees to fp32 summation noise),
  run-to-run determinism, fused infer == forward+vote.
"""
import numpy as np
import pytest

from tests.common import ref_conv2d

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', [(4, 32, 32, 256, 256, 3), (4, 32, 32, 128, 128, 3), (4, 32, 32, 515, 512, 1),
                                  (4, 32, 32, 512, 512, 1), (4, 32, 32, 512, 256, 1), (2, 64, 64, 32, 16, 1),
                                  (3, 16, 16, 64, 64, 3), (40, 2, 2, 64, 64, 3)],
                         ids=lambda c: 'x'.join(map(str, c)))
def test_top_conv_shapes(gpu, case):
    B, H, W, Cin, Cout, k = case
    rng = np.random.default_rng(1)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((B, H, W, Cout)).astype(np.float32)
    y, st = gpu.conv2d(x, w, scale, shift, True, res, want_stats=True)
    yr, raw = ref_conv2d(x, w, scale, shift, True, res)
    assert np.abs(y - yr).max() / np.abs(yr).max() < 2e-5
    np.testing.assert_allclose(st[0], raw.sum((0, 1, 2)), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(st[1], (raw ** 2).sum((0, 1, 2)), rtol=1e-4, atol=1e-3)


@pytest.fixture(scope='module')
def config2(gpu):
    from densereg_amd.data.synthetic import make_crops
    from oracle import net, pose
    from oracle.graph import NetConfig
    cfg = NetConfig(2, 128, 16)
    B = 40
    dm, poses, cfgs, coms, _ = make_crops(B, 'icvl', seed=20240)
    ndm = pose.norm_dm(dm, coms)
    params = net.make_test_params(cfg, ndm[:4], seed=7)
    h = gpu.handle(cfg, B)
    h.load_params(params)
    h.call('dr_finalize_params', gpu.stream)
    return dict(cfg=cfg, B=B, ndm=ndm, poses=poses, cfgs=cfgs, coms=coms, params=params, h=h)


def test_config2_maps_and_xyz_vs_oracle(gpu, config2):
    from oracle import net, pose
    c = config2
    hm, hm3, um = gpu.forward_eval(c['h'], c['ndm'])
    ep = net.forward_eval(c['cfg'], c['params'], c['ndm'])
    for got, key in ((hm, 'hm_outs'), (hm3, 'hm3_outs'), (um, 'um_outs')):
        assert np.abs(got - ep[key][-1]).max() < 5e-4, key
    xyz = gpu.infer(c['h'], c['ndm'], c['cfgs'], c['coms'])
    ref = pose.estimate_pose_mm(ep['hm_outs'][-1], ep['hm3_outs'][-1], ep['um_outs'][-1], c['ndm'], c['cfgs'], c['coms'])
    # BASELINE.json: <= 0.1 mm mean-joint-error delta vs the reference on identical inputs
    e_hip, e_ref = pose.mean_jnt_error(xyz, c['poses']), pose.mean_jnt_error(ref, c['poses'])
    assert abs(e_hip - e_ref) <= 0.1
    assert pose.mean_jnt_error(xyz, ref) <= 0.1
    # vote alone on identical maps: bit for bit.  (Rounds 1-5 allowed 6 joints beyond 2e-3 mm and a 1 mm tail here: the device's expf and
    # numpy's differed by an ulp, and ten mean-shift iterations between two candidate clusters amplify an ulp on knife-edge joints.
    # The kernel weight is now one fixed sequence of IEEE fp32 operations on both sides -- vote.h::vote_exp, oracle/pose.py::exp_f32.)
    xyz_same = gpu.vote(c['h'], hm, hm3, um, c['ndm'], c['cfgs'], c['coms'])
    np.testing.assert_array_equal(xyz_same, pose.estimate_pose_mm(hm, hm3, um, c['ndm'], c['cfgs'], c['coms']))
    np.testing.assert_array_equal(xyz_same, xyz)          # fused infer == forward + vote


def test_config2_batch_invariance_and_determinism(gpu, config2):
    c = config2
    a = gpu.forward_eval(c['h'], c['ndm'])
    b = gpu.forward_eval(c['h'], c['ndm'])
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)               # run-to-run determinism
    # batch-composition invariance: eval mode has no cross-sample term and every output element is one fixed-order
    # fma chain, so a sample's maps do not depend on its neighbours -- bit-exact as long as the batch size selects
    # the same conv tiles (39 of the 40 samples, shuffled) ...
    sel = [37, 5, 18] + [i for i in range(40) if i not in (37, 5, 18, 11)]
    sub = gpu.forward_eval(c['h'], np.ascontiguousarray(c['ndm'][sel]))
    for x, y in zip(a, sub):
        np.testing.assert_array_equal(x[sel], y)
    # ... and to fp32 summation-order noise when it does not (B = 3 runs every layer on the small-grid tiles, which
    # split K over more accumulators)
    sel = [37, 5, 18]
    sub = gpu.forward_eval(c['h'], np.ascontiguousarray(c['ndm'][sel]))
    for x, y in zip(a, sub):
        assert np.abs(x[sel] - y).max() < 2e-5 * max(1.0, float(np.abs(y).max()))


def test_every_layer_config2_small_batch(gpu, config2):
    from oracle import net
    from oracle.graph import conv_specs
    c = config2
    ndm = np.ascontiguousarray(c['ndm'][:2])
    c['h'].call('dr_set_fusion', 0)                     # every layer's output in HBM
    gpu.forward_eval(c['h'], ndm)
    rec = {}
    net.forward_eval(c['cfg'], c['params'], ndm, record=rec)
    for cs in conv_specs(c['cfg']):
        a = gpu.read_activation(c['h'], cs.name, (2, cs.h_out, cs.w_out, cs.cout))
        r = rec.get(cs.name + '+res', rec[cs.name])
        assert np.abs(a - r).max() / (np.abs(r).max() + 1e-12) < 2e-4, cs.name
    c['h'].call('dr_set_fusion', 1)


def test_input_256_maps_64_forward_and_vote(gpu):
    """um_v1.py:99-104 accepts 256x256 crops (hourglass depth 5, 64x64 maps): BASELINE config 5's geometry
    (fp32 here), S=1 F=64 J=14 B=2 against the oracle."""
    from densereg_amd.data.synthetic import make_crops
    from oracle import net, pose
    from oracle.graph import NetConfig
    cfg = NetConfig(1, 64, 14, in_hw=256)
    dm, poses, cfgs, coms, _ = make_crops(2, 'nyu', seed=3, hw=256)
    ndm = pose.norm_dm(dm, coms)
    params = net.make_test_params(cfg, ndm, seed=7)
    h = gpu.handle(cfg, 2)
    h.load_params(params)
    h.call('dr_finalize_params', gpu.stream)
    hm, hm3, um = gpu.forward_eval(h, ndm)
    assert hm.shape == (2, 64, 64, 14) and um.shape == (2, 64, 64, 42)
    ep = net.forward_eval(cfg, params, ndm)
    for got, key in ((hm, 'hm_outs'), (hm3, 'hm3_outs'), (um, 'um_outs')):
        assert np.abs(got - ep[key][-1]).max() < 5e-4, key
    xyz = gpu.infer(h, ndm, cfgs, coms)
    ref = pose.estimate_pose_mm(ep['hm_outs'][-1], ep['hm3_outs'][-1], ep['um_outs'][-1], ndm, cfgs, coms, out_hw=64)
    assert pose.mean_jnt_error(xyz, ref) <= 0.1
    # the same geometry on the bf16 matrix cores (config 5 names a bf16 MFMA conv path): carries the precision's own
    # error and nothing else (tests/test_forward_parity.py::test_network_bf16_precision states the criterion)
    h.call('dr_set_precision', 1)
    h.call('dr_finalize_params', gpu.stream)
    maps16 = gpu.forward_eval(h, ndm)
    ep16 = net.forward_eval(cfg, params, ndm, conv_operands='bf16')
    l2 = lambda a, b: float(np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + 1e-12))
    for got, key in zip(maps16, ('hm_outs', 'hm3_outs', 'um_outs')):
        e_prec = l2(ep16[key][-1], ep[key][-1])
        assert l2(got, ep16[key][-1]) <= e_prec + 1e-5 and l2(got, ep[key][-1]) <= 1.25 * e_prec + 1e-5, key
        assert 1e-4 < e_prec < 0.2, (key, e_prec)
    h.close()


def test_graph_replay_matches_direct_launches(gpu, monkeypatch):
    """dr_infer / dr_forward_eval record their launches into an executable graph on first use and replay it
    afterwards (opt-in, DR_GRAPHS=1): same bits as plain launches, across repeated calls, another batch size, other
    caller buffers (a new key), and a parameter reload in between (the graph reads the repacked weights in place)."""
    from densereg_amd.data.synthetic import make_crops
    from oracle import net, pose
    from oracle.graph import NetConfig
    cfg = NetConfig(1, 32, 16)
    B = 5
    dm, poses, cfgs, coms, _ = make_crops(B, 'icvl', seed=77)
    ndm = pose.norm_dm(dm, coms)
    params = net.make_test_params(cfg, ndm[:2], seed=3)
    params2 = {k: (v * 1.01 if k.endswith('weights') else v) for k, v in params.items()}

    def run(graphs):
        if graphs:
            monkeypatch.setenv('DR_GRAPHS', '1')
        else:
            monkeypatch.delenv('DR_GRAPHS', raising=False)
        h = gpu.handle(cfg, B)
        h.load_params(params)
        h.call('dr_finalize_params', gpu.stream)
        outs = []
        for rep in range(3):
            outs.append(gpu.infer(h, ndm, cfgs, coms))                       # new device buffers each call: new keys
        d = [gpu.dev(np.ascontiguousarray(a, np.float32)) for a in (ndm, cfgs, coms)]
        xyz = gpu.empty((B, 48))
        for rep in range(3):                                                 # same buffers: record once, replay twice
            h.call('dr_infer', B, gpu.ptr(d[0]), gpu.ptr(d[1]), gpu.ptr(d[2]), gpu.ptr(xyz), gpu.stream)
            gpu.sync()
            outs.append(gpu.host(xyz).copy())
        h.call('dr_infer', 2, gpu.ptr(d[0]), gpu.ptr(d[1]), gpu.ptr(d[2]), gpu.ptr(xyz), gpu.stream)   # other batch size
        gpu.sync()
        outs.append(gpu.host(xyz)[:2].copy())
        h.load_params(params2)
        h.call('dr_finalize_params', gpu.stream)
        h.call('dr_infer', B, gpu.ptr(d[0]), gpu.ptr(d[1]), gpu.ptr(d[2]), gpu.ptr(xyz), gpu.stream)   # replays the B=5 graph
        gpu.sync()
        outs.append(gpu.host(xyz).copy())
        outs.append(np.concatenate([m.reshape(B, -1) for m in gpu.forward_eval(h, ndm)], 1))
        h.close()
        return outs

    with_graphs, direct = run(True), run(False)
    for a, b in zip(with_graphs, direct):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(with_graphs[0], with_graphs[5])
    assert np.abs(with_graphs[7] - with_graphs[5]).max() > 0                 # the reloaded weights were really used
