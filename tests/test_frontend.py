"""Input front-end: crop + centre of mass from raw frames, training augmentation.

* known answers for the CPU oracle (``oracle/frontend.py``) derived by hand from the reference's definitions;
* parity of the HIP kernels (``densereg_amd/csrc/frontend.h``) against the oracle through the C ABI, on the
  host-fiber emulator and -- ``-m gpu`` -- on an MI355X, including the reference's frame geometries
  (240x320 ICVL, 480x640 NYU), boxes clipped by the frame border, an all-background frame, boxes given
  explicitly, identity and quarter-turn augmentations and full-size batches.

Tolerances: crops are bilinear blends of depths in mm -> 2e-3 mm absolute (fp32, same op order, contraction
off; the only freedom is cos/sin/division ulps); augmentation is nearest-neighbour resampling -> exact except
where an ulp of sin/cos flips a rounding (<= 0.05 % of the pixels allowed); poses 1e-2 mm.
"""
import numpy as np
import pytest

from oracle import frontend as F

BACKENDS = [pytest.param('emu'), pytest.param('gpu', marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def be(request):
    return request.getfixturevalue(request.param)


# ---------------------------------------------------------------------------------------------
# oracle known answers
# ---------------------------------------------------------------------------------------------
def test_oracle_resize_bilinear_known_values():
    img = np.array([[0, 10], [20, 30]], np.float32)
    out = F.resize_bilinear(img, 4, 4)                      # scale 0.5: src = 0, .5, 1, 1.5 ; hi clamps to 1
    np.testing.assert_allclose(out[0], [0, 5, 10, 10])
    np.testing.assert_allclose(out[1], [10, 15, 20, 20])
    np.testing.assert_allclose(out[3], [20, 25, 30, 30])
    same = F.resize_bilinear(img, 2, 2)
    np.testing.assert_array_equal(same, img)                # scale 1 is the identity
    down = F.resize_bilinear(np.arange(16, dtype=np.float32).reshape(4, 4), 2, 2)
    np.testing.assert_array_equal(down, [[0, 2], [8, 10]])  # scale 2: integer source coordinates, no blending


def test_oracle_crop_box_and_camera():
    cfg = np.array([240.0, 240.0, 160.0, 120.0, 320.0, 240.0], np.float32)
    dm = np.full((240, 320), 800.0, np.float32)
    dm[100:140, 150:200] = 400.0
    # two joints projecting to (u,v) = (160,110) and (190,130) at z = 400
    pose = np.array([0, -10 * 400 / 240, 400, 30 * 400 / 240, 10 * 400 / 240, 400], np.float32)
    crop, _, ncfg = F.crop_from_xyz_pose(dm, pose, cfg, 128, 128, dataset='nyu')
    # box: top = 90, left = 140, bottom = 150, right = 210 -> 60 x 70, square 70, off_h = 5, off_w = 0
    rx = np.float32(70 / 128)
    np.testing.assert_allclose(ncfg, [240 / rx, 240 / rx, (160 - 140 + 0) / rx, (120 - 90 + 5) / rx, 128, 128], rtol=1e-6)
    assert crop.shape == (128, 128)
    assert crop[0, 64] == 0.0                                # the zero padding rows above the box
    assert crop[64, 64] == 400.0                             # the hand
    assert crop[20, 5] == 0.0                                # background 800 >= d_th = 400 + 250 removed
    com = F.center_of_mass(crop, ncfg)
    pos = crop[crop > 0]
    assert 400.0 <= com[2] < 650.0 and abs(com[2] - pos.astype(np.float64).mean()) < 1e-3   # blended edge pixels < d_th stay
    # a square box fully inside a uniform hand: no padding, no blending -> every pixel 400, com depth exactly 400
    dm2 = np.full((240, 320), 800.0, np.float32)
    dm2[70:170, 120:230] = 400.0
    pose2 = np.array([0, -15 * 400 / 240, 400, 30 * 400 / 240, 15 * 400 / 240, 400], np.float32)   # v = 105, 135
    crop2, _, ncfg2 = F.crop_from_xyz_pose(dm2, pose2, cfg, 128, 128, dataset='nyu')
    assert (crop2 == 400.0).all()
    com2 = F.center_of_mass(crop2, ncfg2)
    rx2 = np.float32(70 / 128)
    assert com2[2] == 400.0
    np.testing.assert_allclose(com2[:2], [(64 - (160 - 140) / rx2) * 400 / (240 / rx2), (64 - (120 - 85) / rx2) * 400 / (240 / rx2)],
                               rtol=1e-5)


def test_oracle_crop_or_pad_and_rotate():
    img = np.arange(12, dtype=np.float32).reshape(3, 4)
    np.testing.assert_array_equal(F.crop_or_pad(img, 3, 2), img[:, 1:3])
    padded = F.crop_or_pad(img, 5, 4)
    np.testing.assert_array_equal(padded[1:4], img)
    assert padded[0].sum() == 0 and padded[4].sum() == 0
    sq = np.arange(16, dtype=np.float32).reshape(4, 4) + 1
    np.testing.assert_array_equal(F.rotate_nearest(sq, 0.0), sq)
    r = F.rotate_nearest(sq, np.float32(np.pi / 2))
    assert sorted(r.ravel()) == sorted(sq.ravel())            # a quarter turn of a square permutes the pixels
    assert np.array_equal(r, np.rot90(sq, 1)) or np.array_equal(r, np.rot90(sq, -1))


def test_oracle_data_aug_identity():
    rng = np.random.default_rng(0)
    dm = rng.uniform(0, 500, (32, 32)).astype(np.float32)
    cfg = np.array([300.0, 300.0, 16.0, 16.0, 32.0, 32.0], np.float32)
    com = np.array([1.0, -2.0, 400.0], np.float32)
    pose = np.array([5.0, 6.0, 390.0, -7.0, 2.0, 410.0], np.float32)
    out, p2 = F.data_aug_one(dm, pose, cfg, com, 0.0, 1.0, 1.0)
    np.testing.assert_array_equal(out, dm)
    np.testing.assert_allclose(p2, pose, atol=2e-4)


# ---------------------------------------------------------------------------------------------
# parity through the C ABI
# ---------------------------------------------------------------------------------------------
def _frames(rng, B, H, W, J, fx):
    """Synthetic frames: far background, a hand-sized blob of joints 300-600 mm away, some near the border."""
    cfg = np.tile(np.array([fx, fx, W / 2.0, H / 2.0, W, H], np.float32), (B, 1))
    dms = rng.uniform(700, 1500, (B, H, W)).astype(np.float32)
    dms[rng.uniform(size=dms.shape) < 0.1] = 0.0
    poses = np.zeros((B, 3 * J), np.float32)
    for b in range(B):
        z = rng.uniform(300, 600)
        cu, cv = rng.uniform(0.1 * W, 0.9 * W), rng.uniform(0.1 * H, 0.9 * H)
        if b % 4 == 3:
            cu, cv = rng.choice([3.0, W - 3.0]), rng.choice([5.0, H - 5.0])       # box clipped by the frame
        uv = np.stack([cu + rng.uniform(-45, 45, J), cv + rng.uniform(-45, 45, J)], 1)
        zz = z + rng.uniform(-40, 40, J)
        poses[b] = np.stack([(uv[:, 0] - W / 2.0) * zz / fx, (uv[:, 1] - H / 2.0) * zz / fx, zz], 1).reshape(-1)
        y0, x0 = int(np.clip(cv - 50, 0, H - 1)), int(np.clip(cu - 50, 0, W - 1))
        hand = dms[b, y0:y0 + 100, x0:x0 + 100]
        hand[...] = z + rng.uniform(-60, 60, hand.shape)
    return dms, poses, cfg


def _crop_abi(be, dms, poses, cfgs, icvl, out_hw=128, bbx=None):
    B, H, W = dms.shape
    d = be.dev(dms)
    c = be.dev(cfgs)
    crops, ncfg, com = be.empty((B, out_hw, out_hw)), be.empty((B, 6)), be.empty((B, 3))
    if bbx is None:
        p = be.dev(poses)
        rc = be.lib.dr_crop_from_pose(B, be.ptr(d), H, W, be.ptr(p), poses.shape[1] // 3, be.ptr(c), int(icvl), 20.0, out_hw,
                                      be.ptr(crops), be.ptr(ncfg), be.ptr(com), be.stream)
    else:
        bb = be.dev(bbx)
        rc = be.lib.dr_crop_from_bbx(B, be.ptr(d), H, W, be.ptr(bb), be.ptr(c), out_hw, be.ptr(crops), be.ptr(ncfg), be.ptr(com),
                                     be.stream)
    assert rc == 0, rc
    be.sync()
    return be.host(crops), be.host(ncfg), be.host(com)


@pytest.mark.parametrize('geom', [(240, 320, 16, 241.42, True), (480, 640, 14, 588.03, False)], ids=['icvl', 'nyu'])
def test_crop_from_pose_matches_oracle(be, geom):
    H, W, J, fx, icvl = geom
    rng = np.random.default_rng(H)
    B = 4 if be.name == 'emu' else 12
    dms, poses, cfgs = _frames(rng, B, H, W, J, fx)
    dms[1] = 0.0                                             # an all-background frame: com depth falls back to 200
    crops, ncfg, com = _crop_abi(be, dms, poses, cfgs, icvl)
    for b in range(B):
        rc, _, rcfg = F.crop_from_xyz_pose(dms[b], poses[b], cfgs[b], 128, 128, dataset='icvl' if icvl else 'nyu')
        np.testing.assert_allclose(ncfg[b], rcfg, rtol=1e-6, err_msg='cfg %d' % b)
        assert np.abs(crops[b] - rc).max() < 2e-3, b
        np.testing.assert_allclose(com[b], F.center_of_mass(rc, rcfg), rtol=2e-6, atol=1e-4, err_msg='com %d' % b)
    assert com[1, 2] == 200.0


def test_crop_from_bbx_and_center_of_mass_identity(be):
    rng = np.random.default_rng(3)
    B, H, W = 3, 96, 120
    dms = rng.uniform(0, 900, (B, H, W)).astype(np.float32)
    cfgs = np.tile(np.array([200.0, 210.0, 60.0, 48.0, W, H], np.float32), (B, 1))
    bbx = np.array([[10, 20, 70, 60, 600.0], [0, 0, 96, 120, 1e9], [30.2, 5.9, 64.0, 100.0, 450.0]], np.float32)
    crops, ncfg, com = _crop_abi(be, dms, None, cfgs, False, out_hw=64, bbx=bbx)
    for b in range(B):
        rc, _, rcfg = F.crop_from_bbx(dms[b], None, bbx[b], cfgs[b], 64, 64)
        np.testing.assert_allclose(ncfg[b], rcfg, rtol=1e-6)
        assert np.abs(crops[b] - rc).max() < 2e-3
        np.testing.assert_allclose(com[b], F.center_of_mass(rc, rcfg), rtol=2e-6, atol=1e-4)
    # whole-map box at scale 1 = the identity resize: center_of_mass of an existing crop
    sq = rng.uniform(0, 500, (2, 64, 64)).astype(np.float32)
    sq[sq < 100] = 0
    cf = np.tile(np.array([300.0, 300.0, 30.0, 33.0, 64.0, 64.0], np.float32), (2, 1))
    whole = np.tile(np.array([0, 0, 64, 64, 3e38], np.float32), (2, 1))
    crops, ncfg, com = _crop_abi(be, sq, None, cf, False, out_hw=64, bbx=whole)
    np.testing.assert_array_equal(crops, sq)
    np.testing.assert_array_equal(ncfg, cf)
    for b in range(2):
        np.testing.assert_allclose(com[b], F.center_of_mass(sq[b], cf[b]), rtol=2e-6)


def test_crop_from_bbx_outside_the_frame_and_degenerate(be):
    """A box is data: one that leaves the frame (negative corner, beyond H/W) must not read out of bounds -- the missing
    pixels are background, i.e. the result equals the oracle run on the frame embedded in a larger zero canvas -- and a
    zero-area box leaves a finite crop camera (the reference's crop_to_bounding_box raises on both)."""
    rng = np.random.default_rng(11)
    B, H, W, P = 3, 48, 64, 40
    dms = rng.uniform(200, 900, (B, H, W)).astype(np.float32)
    cfgs = np.tile(np.array([200.0, 210.0, 32.0, 24.0, W, H], np.float32), (B, 1))
    bbx = np.array([[-10, -7, 30, 40, 1e9], [20, 30, H + 15, W + 9, 1e9], [-5, -5, H + 5, W + 5, 600.0]], np.float32)
    crops, ncfg, com = _crop_abi(be, dms, None, cfgs, False, out_hw=32, bbx=bbx)
    assert np.isfinite(crops).all() and np.isfinite(ncfg).all() and np.isfinite(com).all()
    for b in range(B):
        canvas = np.zeros((H + 2 * P, W + 2 * P), np.float32)
        canvas[P:P + H, P:P + W] = dms[b]
        shifted = bbx[b] + np.array([P, P, P, P, 0], np.float32)
        ccfg = cfgs[b] + np.array([0, 0, P, P, 0, 0], np.float32)
        rc, _, rcfg = F.crop_from_bbx(canvas, None, shifted, ccfg, 32, 32)
        assert np.abs(crops[b] - rc).max() < 2e-3, b
        np.testing.assert_allclose(ncfg[b], rcfg, rtol=1e-5, atol=1e-4)
    flat = np.array([[10, 10, 10, 10, 1e9], [20, 5, 12, 3, 1e9]], np.float32)          # no area / inverted
    crops, ncfg, com = _crop_abi(be, dms[:2], None, cfgs[:2], False, out_hw=32, bbx=flat)
    assert np.isfinite(crops).all() and np.isfinite(ncfg).all() and np.isfinite(com).all()
    assert (crops == 0).all() and (com[:, 2] == 200.0).all()


def test_crop_from_bbx_on_the_reference_nyu_boxes(be):
    """tests/golden/nyu_bbx_head.npy = the first 16 rows of the reference's data/nyu_bbx.pkl (data, not code): they pin the
    (top, left, bottom, right, depth threshold) convention of crop_from_bbx (data/preprocess.py:81-129; call site
    data/nyu.py:209-214) -- boxes inside the 480x640 NYU frame, thresholds in the sensor's range -- and are the boxes
    this test crops with: engine vs oracle on synthetic NYU-sized frames."""
    import os
    from tests.common import GOLDEN
    bbx = np.load(os.path.join(GOLDEN, 'nyu_bbx_head.npy'))
    assert bbx.shape == (16, 5) and bbx.dtype == np.float32
    top, left, bottom, right, d_th = bbx.T
    assert (top >= 0).all() and (left >= 0).all() and (bottom > top).all() and (right > left).all()
    assert (bottom <= 480).all() and (right <= 640).all() and ((d_th > 500) & (d_th < 1500)).all()
    B = 4 if be.name == 'emu' else 16
    rng = np.random.default_rng(12)
    H, W = 480, 640
    dms = rng.uniform(600, 1400, (B, H, W)).astype(np.float32)
    dms[rng.uniform(size=dms.shape) < 0.1] = 0.0
    cfgs = np.tile(np.array([588.235, 587.084, 320.0, 240.0, W, H], np.float32), (B, 1))
    crops, ncfg, com = _crop_abi(be, dms, None, cfgs, False, out_hw=128, bbx=np.ascontiguousarray(bbx[:B]))
    for b in range(B):
        rc, _, rcfg = F.crop_from_bbx(dms[b], None, bbx[b], cfgs[b], 128, 128)
        np.testing.assert_allclose(ncfg[b], rcfg, rtol=1e-6)
        assert np.abs(crops[b] - rc).max() < 2e-3
        assert (crops[b] < d_th[b]).all()                           # the box's own threshold removed the background
        np.testing.assert_allclose(com[b], F.center_of_mass(rc, rcfg), rtol=2e-6, atol=1e-4)


def _aug_abi(be, dms, poses, cfgs, coms, draws):
    B, H, W = dms.shape
    args = [be.dev(np.ascontiguousarray(a, np.float32)) for a in (dms, poses, cfgs, coms, draws)]
    out, op = be.empty((B, H, W)), be.empty(poses.shape)
    rc = be.lib.dr_data_aug(B, be.ptr(args[0]), H, W, be.ptr(args[1]), poses.shape[1] // 3, be.ptr(args[2]), be.ptr(args[3]),
                            be.ptr(args[4]), be.ptr(out), be.ptr(op), be.stream)
    assert rc == 0, rc
    be.sync()
    return be.host(out), be.host(op)


def test_data_aug_matches_oracle(be):
    from densereg_amd.data.synthetic import make_crops
    B = 6 if be.name == 'emu' else 40
    dm, poses, cfgs, coms, _ = make_crops(B, 'nyu', seed=11)
    dms = np.ascontiguousarray(dm.reshape(B, 128, 128))
    rng = np.random.default_rng(5)
    draws = F.draw_aug_params(rng, B)
    draws[0] = [0.0, 1.0, 1.0]                               # identity
    draws[1] = [np.pi / 2, 0.9, 1.1]                         # quarter turn, extreme ratios (crop one axis, pad the other)
    draws[2] = [-3.0, 1.1, 0.9]
    out, op = _aug_abi(be, dms, poses, cfgs, coms, draws)
    ref_dm, ref_pose = F.data_aug(dms, poses, cfgs, coms, draws)
    np.testing.assert_array_equal(out[0], dms[0])
    mismatch = (out != ref_dm).mean()
    assert mismatch <= 5e-4, mismatch
    np.testing.assert_allclose(op, ref_pose, atol=1e-2, rtol=0)
    np.testing.assert_allclose(op[0], poses[0], atol=1e-3)


def test_frontend_argument_checks(be):
    x = be.empty((1, 8, 8))
    v = be.empty((8,))
    assert be.lib.dr_crop_from_pose(1, None, 8, 8, be.ptr(v), 2, be.ptr(v), 0, 20.0, 8, be.ptr(x), be.ptr(v), be.ptr(v), None) == -1
    assert be.lib.dr_crop_from_pose(0, be.ptr(x), 8, 8, be.ptr(v), 2, be.ptr(v), 0, 20.0, 8, be.ptr(x), be.ptr(v), be.ptr(v), None) == -1
    assert be.lib.dr_data_aug(1, be.ptr(x), 8, 8, be.ptr(v), 2, be.ptr(v), be.ptr(v), be.ptr(v), be.ptr(x), be.ptr(v), None) == -1   # in place


@pytest.mark.gpu
def test_host_mirror_front_end_to_network_input(gpu):
    """densereg_amd.data.preprocess (torch tensors in, reference function names): frames -> crops/com -> norm_dm."""
    import torch
    from densereg_amd.data import preprocess as P
    rng = np.random.default_rng(9)
    dms, poses, cfgs = _frames(rng, 8, 240, 320, 16, 241.42)
    t = [torch.from_numpy(a).cuda() for a in (dms, poses, cfgs)]
    crops, p2, ncfg, com = P.crop_and_com_from_pose(t[0], t[1], t[2], 128, 128, dataset='icvl')
    torch.cuda.synchronize()
    for b in range(8):
        rc, _, rcfg = F.crop_from_xyz_pose(dms[b], poses[b], cfgs[b], 128, 128, dataset='icvl')
        assert np.abs(crops[b].cpu().numpy() - rc).max() < 2e-3
    com2 = P.center_of_mass(crops, ncfg)
    torch.cuda.synchronize()
    np.testing.assert_allclose(com2.cpu().numpy(), com.cpu().numpy(), rtol=1e-6)
    draws = torch.from_numpy(F.draw_aug_params(rng, 8)).cuda()
    aug, ap = P.data_aug(crops, p2, ncfg, com, draws)
    torch.cuda.synchronize()
    ref_dm, ref_pose = F.data_aug(crops.cpu().numpy(), poses, ncfg.cpu().numpy(), com.cpu().numpy(), draws.cpu().numpy())
    assert (aug.cpu().numpy() != ref_dm).mean() <= 5e-4
    np.testing.assert_allclose(ap.cpu().numpy(), ref_pose, atol=1e-2)
