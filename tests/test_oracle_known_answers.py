"""Hand-derived known-answer tests that pin the closed-form pieces of the CPU oracle.

The reference has no tests and cannot run here: these cases are worked out by
hand from the cited reference lines and the TF-semantics checklist (asymmetric SAME padding, max-pool
ignoring padding, bicubic/4 == [::4, ::4], NN upsample floor(i/2), biased variance, eps inside sqrt,
l2_loss = sum/2, Adam eps outside the bias-corrected sqrt, top_k ties -> lower index, tf.where
row-major [-1], truncating float->int casts, gather_nd OOB -> 0).
"""
import math

import numpy as np
import torch

from oracle import net, pose, train
from oracle.graph import NetConfig, SpecOps, conv_specs, param_specs, same_pad, trainable_names

f32 = np.float32


def test_same_padding_rule():
    assert same_pad(128, 7, 2) == (2, 3)      # stem conv 7x7/s2 @128
    assert same_pad(32, 3, 1) == (1, 1)
    assert same_pad(32, 3, 2) == (0, 1)       # hourglass pool 3x3/s2 on even H: pad bottom/right only
    assert same_pad(64, 2, 2) == (0, 0)
    assert same_pad(2, 3, 2) == (0, 1)


def test_layer_inventory_matches_survey():
    cs = conv_specs(NetConfig(2, 128, 16))
    assert len(cs) == 146 and sum(c.bn for c in cs) == 134
    assert abs(sum(c.flops_per_crop for c in cs) / 1e9 - 9.790) < 1e-3
    assert len(trainable_names(NetConfig(2, 128, 16))) == 426
    assert sum(int(np.prod(s)) for _, s, t in param_specs(NetConfig(2, 128, 16)) if t) == 5856352
    names = [c.name for c in cs]
    assert names[0] == 'hg_imgproc/Conv' and names[11] == 'hg_imgproc/Conv_11' and names[12] == 'Conv'
    assert (cs[12 + 40].name, cs[12 + 40].cout, cs[12 + 40].bn) == ('Conv_40', 16, False)      # hm_out
    assert (cs[12 + 63].cin, cs[12 + 63].cout) == (515, 512)                                     # um_full 1
    assert cs[-1].name == 'Conv_133' and cs[-1].cout == 48
    for J, gf in ((14, 9.732), (21, 9.939)):
        assert abs(sum(c.flops_per_crop for c in conv_specs(NetConfig(2, 128, J))) / 1e9 - gf) < 1e-3
    assert abs(sum(c.flops_per_crop for c in conv_specs(NetConfig(1, 64, 16))) / 1e9 - 4.262) < 1e-3
    assert abs(sum(c.flops_per_crop for c in conv_specs(NetConfig(4, 256, 14, 256))) / 1e9 - 115.07) < 1e-2


def _ops(is_training=False):
    return net.TorchOps(NetConfig(1, 8, 2), {}, is_training)


def test_max_pool_same_ignores_padding():
    x = -torch.arange(16, dtype=torch.float32).view(1, 1, 4, 4) - 1.0       # all negative
    y = _ops().max_pool(x, 3, 2)
    # window rows {0,1,2} / {2,3,(pad)}: zero padding would have won everywhere
    assert y.view(-1).tolist() == [-1.0, -3.0, -9.0, -11.0]
    y2 = _ops().max_pool(x, 2, 2)
    assert y2.view(-1).tolist() == [-1.0, -3.0, -9.0, -11.0]


def test_upsample_nearest_and_tiny_dm_and_uvd():
    o = _ops()
    x = torch.tensor([[1.0, 2.0], [3.0, 4.0]]).view(1, 1, 2, 2)
    assert o.upsample2(x).view(4, 4).tolist() == [[1, 1, 2, 2], [1, 1, 2, 2], [3, 3, 4, 4], [3, 3, 4, 4]]
    dm = torch.arange(64, dtype=torch.float32).view(1, 1, 8, 8)
    assert o.tiny_dm(dm).reshape(-1).tolist() == [0.0, 4.0, 32.0, 36.0]
    uvd = o.uvd(torch.zeros(1, 1, 4, 4))
    assert uvd[0, 0, 2].tolist() == [-1.0, -0.5, 0.0, 0.5]       # uu[i,j] = j/(w/2)-1
    assert uvd[0, 1, :, 1].tolist() == [-1.0, -0.5, 0.0, 0.5]    # vv[i,j] = i/(h/2)-1


def test_conv_same_asymmetric_padding_stride2():
    # 1 channel ones image, 7x7 ones kernel, stride 2 on 8x8: out[0,0] sees rows/cols -2..4 -> 5x5 valid
    p = {'Conv/weights': torch.ones(7, 7, 1, 1), 'Conv/biases': torch.zeros(1)}
    o = net.TorchOps(NetConfig(1, 8, 2), p, False)
    y = o.conv(torch.ones(1, 1, 8, 8), 1, 7, 2, bn=False, relu=False, wd=0)
    # total pad = (4-1)*2+7-8 = 5 -> (2,3)
    assert y.shape == (1, 1, 4, 4)
    assert y[0, 0, 0, 0].item() == 25.0 and y[0, 0, 3, 3].item() == 16.0 and y[0, 0, 1, 1].item() == 49.0


def test_batch_renorm_eval_and_train_closed_form():
    C = 2
    base = 'Conv/BatchReNorm/'
    p = {'Conv/weights': torch.eye(C).view(1, 1, C, C), base + 'beta': torch.tensor([0.5, -1.0]),
         base + 'gamma': torch.tensor([2.0, 3.0]), base + 'moving_mean': torch.tensor([1.0, 0.0]),
         base + 'moving_variance': torch.tensor([3.999, 0.999]), base + 'r_max': torch.tensor([1.5]),
         base + 'd_max': torch.tensor([0.25]), base + 'curr_t': torch.tensor([0.0])}
    x = torch.tensor([[1.0, 2.0], [3.0, 6.0]]).view(2, C, 1, 1)      # per-channel values (1,3) and (2,6)
    # eval: gamma*(x-mm)/sqrt(mv+eps)+beta, eps=1e-3 inside the sqrt
    y = net.TorchOps(NetConfig(1, 8, 2), p, False).conv(x, C, 1, 1, bn=True, relu=False, wd=0)
    np.testing.assert_allclose(y.view(2, C).numpy(), [[0.5, 5.0], [2.5, 17.0]], rtol=1e-6)
    # train: mean=(2,4), biased var=(1,4), std=sqrt(var+1e-3); r=clip(std/mstd,1/1.5,1.5), d=clip((mean-mm)/mstd,+-.25)
    ops = net.TorchOps(NetConfig(1, 8, 2), p, True)
    y = ops.conv(x, C, 1, 1, bn=True, relu=False, wd=0)
    std = np.sqrt(np.array([1.0, 4.0]) + 1e-3)
    r = np.clip(std / np.array([2.0, 1.0]), 1 / 1.5, 1.5)
    d = np.clip((np.array([2.0, 4.0]) - np.array([1.0, 0.0])) / np.array([2.0, 1.0]), -0.25, 0.25)
    xhat = (np.array([[1.0, 2.0], [3.0, 6.0]]) - np.array([2.0, 4.0])) / std
    exp = (xhat * r + d) * np.array([2.0, 3.0]) + np.array([0.5, -1.0])
    np.testing.assert_allclose(y.view(2, C).detach().numpy(), exp, rtol=1e-5)
    assert np.allclose(r, [2 / 3, 1.5]) and np.allclose(d, [0.25, 0.25])       # both clips active
    u = ops.bn_updates['Conv']
    np.testing.assert_allclose(u['mean'].numpy(), [2.0, 4.0])
    np.testing.assert_allclose(u['var'].numpy(), [1.0, 4.0])


def test_bn_state_update_zero_debias_and_schedule():
    base = 'Conv/BatchReNorm/'
    params = {base + 'moving_mean': np.zeros(1, f32), base + 'moving_variance': np.ones(1, f32),
              base + 'r_max': np.ones(1, f32), base + 'd_max': np.zeros(1, f32), base + 'curr_t': np.zeros(1, f32)}
    upd = {'Conv': {'mean': torch.tensor([5.0]), 'var': torch.tensor([2.0])}}
    shadow = {}
    net.bn_state_update(params, upd, zero_debias=True, shadow=shadow)
    # first debiased average equals the first value exactly: 0.01*5 / (1-0.99)
    assert abs(params[base + 'moving_mean'][0] - 5.0) < 1e-5 and abs(params[base + 'moving_variance'][0] - 2.0) < 1e-5
    assert abs(params[base + 'r_max'][0] - 1.0) < 1e-7                 # 3/(1+2e^0)
    assert abs(params[base + 'd_max'][0] - 1e-3) < 1e-9                # 5/(5000 e^0)
    assert abs(params[base + 'curr_t'][0] - 1e-5) < 1e-12
    p2 = dict(params)
    p2[base + 'moving_mean'] = np.zeros(1, f32)
    net.bn_state_update(p2, upd, zero_debias=False)
    assert abs(p2[base + 'moving_mean'][0] - 0.05) < 1e-7              # plain EMA: 0 + 0.01*(5-0)


def test_norm_dm_known_values():
    com = np.array([[0.0, 0.0, 500.0]], f32)
    dm = np.array([0.0, 199.9, 200.1, 350.0, 500.0, 649.9, 650.0, 700.0], f32).reshape(1, 1, 8, 1)
    out = pose.norm_dm(dm, com).reshape(-1)
    # valid iff 200 < d < 650 ; value (d-350)/300
    np.testing.assert_allclose(out, [-1, -1, (200.1 - 350) / 300, 0.0, 0.5, (649.9 - 350) / 300, -1, -1], rtol=1e-5)


def test_point_cloud_and_projection_roundtrip():
    cfg = np.array([[240.0, 200.0, 64.0, 60.0, 128.0, 128.0]], f32)    # /4 -> fx 60, fy 50, cx 16, cy 15
    com = np.array([[10.0, -20.0, 400.0]], f32)
    dm = np.full((1, 32, 32, 1), 0.5, f32)                               # z = 0.5*300 + 250 = 400
    dm[0, 3, 7, 0] = -1.0                                                # background -> z = com_z + 150
    xyz = pose.generate_xyzs(dm, cfg, com)
    np.testing.assert_allclose(xyz[0, 15, 16], [(0.0 - 10) / 100, (0.0 + 20) / 100, 0.0], atol=1e-6)
    np.testing.assert_allclose(xyz[0, 15, 22], [((22 - 16) * 400 / 60 - 10) / 100, 0.2, 0.0], rtol=1e-5)
    np.testing.assert_allclose(xyz[0, 3, 7, 2], 1.5, rtol=1e-6)
    uvd = pose.xyz2uvd(np.array([40.0, -16.0, 400.0], f32), (f32(60), f32(50), f32(16), f32(15)))
    np.testing.assert_allclose(uvd[0], [22.0, 13.0, 400.0], rtol=1e-6)


def test_gt_synthesis_cone_ball_unit_vector():
    cfg = np.array([[240.0, 240.0, 64.0, 64.0, 128.0, 128.0]], f32)     # map camera fx 60, c 16
    com = np.array([[0.0, 0.0, 400.0]], f32)
    pose_mm = np.array([[0.0, 0.0, 400.0, 40.0, 0.0, 400.0]], f32)       # joint0 -> (16,16); joint1 -> u=22
    hm = pose.hm_2d(pose_mm, cfg, 32, 32)
    assert hm[0, 16, 16, 0] == 1.0 and abs(hm[0, 16, 18, 0] - 0.5) < 1e-6 and hm[0, 16, 20, 0] == 0.0
    assert abs(hm[0, 13, 22, 1] - 0.25) < 1e-6                            # 3 px away -> (4-3)/4
    dm = np.full((1, 128, 128, 1), 0.5, f32)
    _, hm3, um = pose.make_targets(dm, pose_mm, cfg, com, 32)
    # pixel (16,19): x = 3*400/60 = 20 mm -> offset to joint0 = (-0.2,0,0) -> hm3 = (0.8-0.2)/0.8, um = (-1,0,0)
    assert abs(hm3[0, 16, 19, 0] - 0.75) < 1e-5
    np.testing.assert_allclose(um[0, 16, 19, 0:3], [-1.0, 0.0, 0.0], atol=1e-5)
    # pixel (16,28): 80 mm away -> d = 0.8 -> hm3 = 0 and the unit vector is cut (d >= 0.79)
    assert hm3[0, 16, 28, 0] == 0.0 and np.all(um[0, 16, 28, 0:3] == 0.0)


def test_top_k_ties_lower_index_first():
    v = np.array([0.0, 3.0, 1.0, 3.0, 3.0, 0.5, 1.0], f32)
    assert pose.top_k_indices(v, 5).tolist() == [1, 3, 4, 2, 6]


def test_candidate_weight_truncation_and_oob():
    hm = np.arange(32 * 32, dtype=f32).reshape(32, 32)
    com = np.zeros(3, f32)
    cfg4 = (f32(100), f32(100), f32(16), f32(16))
    # p*100+com -> (x,y,z) mm ; u = x*100/z+16
    w = pose.candidate_weight(np.array([0.0, 0.0, 1.0], f32), com, cfg4, hm)            # u=v=16 -> +0.5 -> 16
    assert w == hm[16, 16]
    w = pose.candidate_weight(np.array([-0.164, 0.0, 1.0], f32), com, cfg4, hm)         # u=-0.4 -> +0.5=0.1 -> 0
    assert w == hm[16, 0]
    w = pose.candidate_weight(np.array([-0.169, 0.0, 1.0], f32), com, cfg4, hm)         # u+0.5 = -0.4 -> int 0
    assert w == hm[16, 0]
    assert pose.candidate_weight(np.array([-0.2, 0.0, 1.0], f32), com, cfg4, hm) == 0   # u+0.5=-3.5 -> OOB -> 0
    assert pose.candidate_weight(np.array([0.16, 0.0, 1.0], f32), com, cfg4, hm) == 0   # u+0.5=32.5 -> OOB
    assert pose.candidate_weight(np.array([0.0, 0.0, 0.0], f32), com, cfg4, hm) == 0    # 0/0 -> NaN -> 0


def test_mean_shift_start_cell_and_fixed_point():
    can = np.array([[0.1, 0.1, 0.1]] * 5, f32)
    c = pose.weighted_mean_shift(can, np.ones(5, f32))
    np.testing.assert_allclose(c, [0.1, 0.1, 0.1], rtol=1e-6)       # all candidates equal -> that point
    # two clusters: weights decide the start cell; ties between cells -> LAST cell in row-major order
    can = np.array([[-0.9, -0.9, -0.9], [-0.9, -0.9, -0.9], [0.9, 0.9, 0.9], [0.9, 0.9, 0.9], [0.9, 0.9, 0.9]], f32)
    c = pose.weighted_mean_shift(can, np.array([1.5, 1.5, 1.0, 1.0, 1.0], f32))          # 3.0 vs 3.0 -> last cell
    assert np.all(c > 0.8)
    c = pose.weighted_mean_shift(can, np.array([2.0, 2.0, 1.0, 1.0, 1.0], f32))
    assert np.all(c < -0.8)
    # all weights negative: max of the histogram is 0 at an EMPTY cell; last empty cell = (3,3,2) -> centre (.75,.75,.25)
    c = pose.weighted_mean_shift(can, -np.ones(5, f32), num_it=0)
    np.testing.assert_allclose(c, [0.75, 0.75, 0.25])
    # zero mass: guard keeps the start centre
    c = pose.weighted_mean_shift(can, np.zeros(5, f32))
    np.testing.assert_allclose(c, [0.75, 0.75, 0.75])


def test_l2_and_reg_loss_definition():
    cfg = NetConfig(1, 8, 2)
    params = net.init_params(cfg, 1)
    tp = net.to_torch_params(params)
    reg = float(net.reg_loss(cfg, tp))
    exp = sum(0.0005 * 0.5 * float((params[c.name + '/weights'].astype(np.float64) ** 2).sum())
              for c in conv_specs(cfg) if c.weight_decay > 0)
    assert abs(reg - exp) / exp < 1e-5
    assert sum(1 for c in conv_specs(NetConfig(2, 8, 2)) if c.weight_decay == 0) == 2     # re-inject convs only


def test_adam_step_tf_rule():
    p = {'w': np.array([1.0, -2.0], f32)}
    m = {'w': np.zeros(2, f32)}
    v = {'w': np.zeros(2, f32)}
    acc = {'w': np.array([0.5, -5.0], f32)}          # /5 -> (0.1, -1.0) -> clip -> (0.1, -0.2)
    train.adam_step(p, m, v, acc, lr=1e-3, t=1, div=5.0)
    g = np.array([0.1, -0.2])
    mm, vv = 0.5 * g, 0.001 * g * g
    lr_t = 1e-3 * math.sqrt(1 - 0.999) / (1 - 0.5)
    np.testing.assert_allclose(p['w'], np.array([1.0, -2.0]) - lr_t * mm / (np.sqrt(vv) + 1e-8), rtol=1e-6)
    assert train.learning_rate(0, 1e-3, 100.5) == 1e-3 and abs(train.learning_rate(101, 1e-3, 100.5) - 1e-4) < 1e-12


def test_evaluation_metrics():
    a = np.zeros(6, f32)
    b = np.array([3, 4, 0, 0, 0, 12], f32)
    assert pose.mean_jnt_error(a, b) == 8.5 and pose.max_jnt_error(a, b) == 12.0
