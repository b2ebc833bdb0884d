"""Dataset formats in front of the crop front-end: PNG depth frames, TFRecord / tf.train.Example,
the three dataset adapters.  Pins: Pillow as the independent PNG codec, the RFC 3720 CRC-32C check value, a
hand-assembled Example message; the engine's C / HIP pieces against ``oracle/dataio.py``."""
import io
import os
import struct

import numpy as np
import pytest

from densereg_amd.data import datasets, png, tfrecord
from oracle import dataio as oracle_io

BACKENDS = [pytest.param('emu'), pytest.param('gpu', marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def be(request):
    return request.getfixturevalue(request.param)


def _depth_frame(rng, h, w, base=600):
    yy, xx = np.mgrid[0:h, 0:w]
    d = base + 80 * np.sin(xx / 9.0) * np.cos(yy / 11.0) + rng.normal(0, 3, (h, w))
    d[(yy - h / 2) ** 2 + (xx - w / 2) ** 2 > (min(h, w) * 0.35) ** 2] = 0          # background
    return np.clip(d, 0, 65535).astype(np.uint16)


def _nyu_rgb(depth16, rng):
    rgb = np.zeros(depth16.shape + (3,), np.uint8)
    rgb[..., 0] = rng.integers(0, 256, depth16.shape)            # R carries the synthetic-hand mask in NYU: ignored
    rgb[..., 1] = depth16 >> 8
    rgb[..., 2] = depth16 & 0xFF
    return rgb


def _pil_png(a, **kw):
    from PIL import Image
    buf = io.BytesIO()
    (Image.fromarray(a) if a.dtype == np.uint8 else Image.fromarray(a.astype(np.uint16))).save(buf, format='PNG', **kw)
    return buf.getvalue()


# ---------------------------------------------------------------------------------------------
# oracle pins
# ---------------------------------------------------------------------------------------------
def test_oracle_crc32c_and_record_mask_known_answers():
    assert oracle_io.crc32c(b'123456789') == 0xE3069283                                # RFC 3720 B.4
    assert oracle_io.crc32c(bytes(32)) == 0x8A9136AA
    c = 0xE3069283
    assert oracle_io.masked_crc(b'123456789') == (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def test_oracle_png_decoder_against_pillow():
    from PIL import Image
    rng = np.random.default_rng(0)
    d16 = _depth_frame(rng, 24, 40)
    for blob, ref in ((_pil_png(d16), d16), (_pil_png(d16, compress_level=1), d16), (_pil_png(d16, optimize=True), d16)):
        w, h, depth, ch, s = oracle_io.png_decode(blob)
        assert (w, h, depth, ch) == (40, 24, 16, 1)
        np.testing.assert_array_equal(oracle_io.depth_from_samples(s, ch, depth), ref.astype(np.float32))
    rgb = _nyu_rgb(d16, rng)
    w, h, depth, ch, s = oracle_io.png_decode(_pil_png(rgb))
    assert (depth, ch) == (8, 3)
    np.testing.assert_array_equal(s.reshape(24, 40, 3), rgb)
    np.testing.assert_array_equal(oracle_io.depth_from_samples(s, ch, depth), d16.astype(np.float32))
    # Pillow also reads back what the oracle's inverse understands of every filter type
    for ft in range(5):
        img = Image.open(io.BytesIO(png.encode_png(d16, filter_type=ft)))
        np.testing.assert_array_equal(np.asarray(img).astype(np.uint16), d16)
        img = Image.open(io.BytesIO(png.encode_png(rgb, filter_type=ft)))
        np.testing.assert_array_equal(np.asarray(img), rgb)


def test_oracle_example_wire_format_known_answer():
    # Example{features{feature{"a": float_list{1.0, 2.0}}}} assembled by hand from the protobuf encoding rules
    flist = b'\x0a\x08' + struct.pack('<ff', 1.0, 2.0)              # FloatList.value, packed
    feat = b'\x12' + bytes([len(flist)]) + flist                   # Feature.float_list
    entry = b'\x0a\x01a' + b'\x12' + bytes([len(feat)]) + feat     # map entry: key, value
    features = b'\x0a' + bytes([len(entry)]) + entry
    msg = b'\x0a' + bytes([len(features)]) + features
    assert tfrecord.make_example({'a': np.array([1.0, 2.0], np.float32)}) == msg
    np.testing.assert_array_equal(oracle_io.example_features(msg)['a'], [1.0, 2.0])
    np.testing.assert_array_equal(tfrecord.parse_example(msg)['a'], [1.0, 2.0])
    # unpacked floats (one fixed32 per value) parse too
    unpacked = b'\x0d' + struct.pack('<f', 3.0) + b'\x0d' + struct.pack('<f', 4.0)
    feat = b'\x12' + bytes([len(unpacked)]) + unpacked
    entry = b'\x0a\x01b' + b'\x12' + bytes([len(feat)]) + feat
    features = b'\x0a' + bytes([len(entry)]) + entry
    np.testing.assert_array_equal(tfrecord.parse_example(b'\x0a' + bytes([len(features)]) + features)['b'], [3.0, 4.0])


def test_committed_fixtures_pillow_png_and_record_shard(tmp_path):
    """tests/golden/dataio_fixtures.npz (made by tests/golden/make_dataio_fixtures.py): PNG streams written by Pillow and
    the sample arrays they hold; a TFRecord shard with two records.  Engine decoder and oracle agree with the data."""
    from tests.common import golden
    g = golden('dataio_fixtures.npz')
    for key, ref, ch, depth in (('png_grey16', g['depth16'], 1, 16), ('png_grey16_opt', g['depth16'], 1, 16), ('png_rgb8', g['rgb'], 3, 8)):
        blob = g[key].tobytes()
        info, s = png.decode_png(blob)
        assert (info.width, info.height, info.channels, info.bit_depth) == (40, 24, ch, depth)
        np.testing.assert_array_equal(s, oracle_io.png_decode(blob)[4])
        np.testing.assert_array_equal(oracle_io.depth_from_samples(s, ch, depth), g['depth16'].astype(np.float32))
        if ch == 3:
            np.testing.assert_array_equal(s.reshape(24, 40, 3), ref)
    path = str(tmp_path / 'shard')
    open(path, 'wb').write(g['shard'].tobytes())
    recs = list(tfrecord.read_records(path))
    assert len(recs) == 2 and recs[1] == b'second' and oracle_io.records(g['shard'].tobytes()) == recs
    f = tfrecord.parse_example(recs[0])
    assert f['name'] == [b'test_seq_1/image_0000.png'] and f['png16'][0] == g['png_grey16'].tobytes()
    np.testing.assert_array_equal(f['xyz_pose'], g['pose'])


# ---------------------------------------------------------------------------------------------
# host pieces of the engine
# ---------------------------------------------------------------------------------------------
def test_png_decode_every_filter_and_pillow_files():
    rng = np.random.default_rng(1)
    d16 = _depth_frame(rng, 31, 53)                                                    # odd sizes
    rgb = _nyu_rgb(d16, rng)
    noise16 = rng.integers(0, 65536, (17, 9)).astype(np.uint16)                        # worst case for the predictors
    for img in (d16, rgb, noise16):
        blobs = [png.encode_png(img, filter_type=ft) for ft in range(5)] + [_pil_png(img), _pil_png(img, optimize=True)]
        for blob in blobs:
            info, s = png.decode_png(blob)
            w, h, depth, ch, so = oracle_io.png_decode(blob)
            assert (info.width, info.height, info.bit_depth, info.channels) == (w, h, depth, ch)
            np.testing.assert_array_equal(s, so)
            np.testing.assert_array_equal(oracle_io.depth_from_samples(s, ch, depth).reshape(img.shape[:2]),
                                          (img if img.ndim == 2 else d16).astype(np.float32))
    with pytest.raises(png.PngError):
        png.decode_png(b'not a png at all')
    bad = bytearray(png.encode_png(d16)); bad[40] ^= 0xFF
    with pytest.raises(png.PngError):
        png.decode_png(bytes(bad))                                                     # chunk CRC
    with pytest.raises(png.PngError):
        png.decode_png(_pil_png(np.zeros((4, 4, 4), np.uint8)))                        # RGBA: not a depth frame


def test_png_unfilter_rejects_unknown_filter_type():
    from densereg_amd import _lib
    lib = _lib.load()
    raw = np.zeros((2, 9), np.uint8); raw[1, 0] = 5
    out = np.zeros((2, 8), np.uint8)
    assert lib.dr_png_unfilter(raw.ctypes.data, 2, 8, 2, out.ctypes.data) != 0
    assert lib.dr_png_unfilter(None, 2, 8, 2, out.ctypes.data) != 0


def test_tfrecord_round_trip_and_corruption(tmp_path):
    rng = np.random.default_rng(2)
    recs = [tfrecord.make_example({'name': ('frame_%d.png' % i).encode(), 'xyz_pose': rng.standard_normal(48).astype(np.float32),
                                   'png16': rng.integers(0, 256, 5000 + i).astype(np.uint8).tobytes(), 'idx': np.array([i, -i])})
            for i in range(5)] + [b'']
    path = str(tmp_path / 'testing-0-of-1')
    assert tfrecord.write_records(path, recs) == 6
    back = list(tfrecord.read_records(path))
    assert back == recs
    assert oracle_io.records(open(path, 'rb').read()) == recs                          # bitwise CRC restatement agrees
    f = tfrecord.parse_example(back[3])
    assert f['name'] == [b'frame_3.png'] and f['xyz_pose'].shape == (48,) and len(f['png16'][0]) == 5003
    np.testing.assert_array_equal(f['idx'], [3, -3])
    o = oracle_io.example_features(back[3])
    np.testing.assert_array_equal(o['xyz_pose'], f['xyz_pose'])
    assert o['png16'] == f['png16']
    blob = bytearray(open(path, 'rb').read())
    blob[30] ^= 1
    open(path, 'wb').write(bytes(blob))
    with pytest.raises(tfrecord.RecordError):
        list(tfrecord.read_records(path))
    assert len(list(tfrecord.read_records(path, verify=False))) == 6
    open(path, 'wb').write(bytes(blob[:-3]))
    with pytest.raises(tfrecord.RecordError):
        list(tfrecord.read_records(path, verify=False))


# ---------------------------------------------------------------------------------------------
# device step
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('npix', [1, 3, 4, 5, 1023, 320 * 240 + 2])
def test_depth_from_samples_kernel(be, npix):
    rng = np.random.default_rng(npix)
    for mode, per in ((0, 3), (1, 2)):
        s = rng.integers(0, 256, npix * per).astype(np.uint8)
        d_s = be.dev(np.concatenate([s, np.zeros((-len(s)) % 4, np.uint8)]))           # any 4-byte aligned buffer
        out = be.empty((npix + 3,), np.float32)
        d_out = be.dev(np.full(npix + 3, -7.0, np.float32))
        assert be.lib.dr_depth_from_samples(be.ptr(d_s), npix, mode, be.ptr(d_out), be.stream) == 0
        be.sync()
        got = be.host(d_out)
        ref = oracle_io.depth_from_samples(s.reshape(1, -1), 3 if mode == 0 else 1, 8 if mode == 0 else 16).reshape(-1)
        np.testing.assert_array_equal(got[:npix], ref)
        assert np.all(got[npix:] == -7.0)                                                # nothing past the last pixel
        del out
    assert be.lib.dr_depth_from_samples(be.ptr(d_s), npix, 2, be.ptr(d_out), be.stream) != 0
    assert be.lib.dr_depth_from_samples(None, npix, 0, be.ptr(d_out), be.stream) != 0


# ---------------------------------------------------------------------------------------------
# dataset adapters
# ---------------------------------------------------------------------------------------------
def _make_icvl(root, rng, n=7):
    d = os.path.join(root, 'Testing', 'Depth', 'test_seq_1')
    os.makedirs(d)
    cfg = datasets.IcvlDataset.cfg
    frames, lines = [], []
    for i in range(n):
        dm = _depth_frame(rng, cfg.h, cfg.w, base=300 + 10 * i)
        frames.append(dm)
        open(os.path.join(d, 'image_%04d.png' % i), 'wb').write(_pil_png(dm))
        fg = np.argwhere(dm > 0)
        pts = fg[rng.integers(0, len(fg), 16)]
        uvd = np.stack([pts[:, 1], pts[:, 0], dm[pts[:, 0], pts[:, 1]]], 1).astype(np.float64)
        lines.append('test_seq_1/image_%04d.png %s\n' % (i, ' '.join('%.4f' % v for v in uvd.reshape(-1))))
    open(os.path.join(root, 'Testing', 'labels.txt'), 'w').writelines(lines)
    return frames


def test_icvl_adapter_annotations_shards_and_parse(tmp_path):
    rng = np.random.default_rng(3)
    root = str(tmp_path)
    frames = _make_icvl(root, rng)
    ds = datasets.IcvlDataset('testing', root)
    ann = ds.loadAnnotation()
    assert len(ann) == 7 and ann[0].name == 'test_seq_1/image_0000.png' and len(ann[0].pose) == 48
    # uvd -> xyz -> uvd round trip of the label line (data/util.py:20-21)
    first = np.array([float(v) for v in open(os.path.join(root, 'Testing', 'labels.txt')).readline().split()[1:]])
    np.testing.assert_allclose(datasets.xyz2uvd(np.array(ann[0].pose), ds.cfg).reshape(-1), first, rtol=1e-9, atol=1e-6)
    paths = ds.write_TFRecord(num_shards=4, num_threads=2)
    assert [os.path.basename(p) for p in paths] == ['testing-%d-of-4' % i for i in range(4)]
    assert ds.filenames[:4] == paths and ds.filenames[4] == paths[3]                   # the repeated last shard (icvl.py:73-74)
    counts = [len(list(tfrecord.read_records(p))) for p in paths]
    assert counts == [1, 2, 2, 2]                                                       # np.linspace boundaries of the reference
    k = 0
    for rec in ds.records(shuffle=False, epochs=1, files=paths):
        info, samples, pose, name, bbx = ds.parse_example(rec)
        assert name == ann[k].name and bbx is None
        np.testing.assert_allclose(pose, np.asarray(ann[k].pose, np.float32))
        np.testing.assert_array_equal(oracle_io.depth_from_samples(samples, 1, 16).reshape(240, 320), frames[k].astype(np.float32))
        k += 1
    assert k == 7
    # shuffled reading visits every record exactly once per epoch; ranks read disjoint shards
    seen = sorted(tfrecord.parse_example(r)['name'][0] for r in ds.records(shuffle=True, seed=5, epochs=1, files=paths))
    assert seen == sorted(a.name.encode() for a in ann)
    r0 = [tfrecord.parse_example(r)['name'][0] for r in ds.records(False, epochs=1, files=paths, rank=0, world=2)]
    r1 = [tfrecord.parse_example(r)['name'][0] for r in ds.records(False, epochs=1, files=paths, rank=1, world=2)]
    assert len(r0) + len(r1) == 7 and not set(r0) & set(r1)
    with pytest.raises(ValueError):
        datasets.IcvlDataset('nonsense', root)
    # the host half of the batch pipeline: prefetching producer thread and decode pool give the stream of the plain loop
    plain = [[it[3] for it in b] for b in ds.host_batches(3, False, epochs=1, files=paths, workers=1, prefetch=0)]
    ahead = [[it[3] for it in b] for b in ds.host_batches(3, False, epochs=1, files=paths, workers=4, prefetch=2)]
    assert plain == ahead and [len(b) for b in plain] == [3, 3, 1] and sum(plain, []) == [a.name for a in ann]
    assert [len(b) for b in ds.host_batches(3, False, epochs=1, files=paths, drop_last=True)] == [3, 3]
    it = ds.host_batches(2, True, seed=1, epochs=None, files=paths)            # endless stream, abandoned early: no hang
    assert len(next(it)) == 2 and len(next(it)) == 2
    it.close()
    blob = bytearray(open(paths[1], 'rb').read()); blob[40] ^= 0xFF
    bad = str(tmp_path / 'corrupt-shard')
    open(bad, 'wb').write(bytes(blob))
    with pytest.raises(tfrecord.RecordError):                                   # a producer-side error reaches the consumer
        list(ds.host_batches(2, False, epochs=1, files=[paths[0], bad]))


def test_msra_bin_reader_and_label_signs(tmp_path):
    rng = np.random.default_rng(4)
    root = str(tmp_path)
    g = os.path.join(root, 'P3', '1')
    os.makedirs(g)
    cfg = datasets.MsraDataset.cfg
    full = _depth_frame(rng, cfg.h, cfg.w, base=400).astype(np.float32)
    top, bottom, left, right = 40, 200, 60, 260
    for i, crop in enumerate((full[top:bottom, left:right], np.zeros((bottom - top, right - left), np.float32))):
        with open(os.path.join(g, '%06d_depth.bin' % i), 'wb') as f:
            f.write(struct.pack('<6i', cfg.w, cfg.h, left, top, right, bottom))
            f.write(crop.astype('<f4').tobytes())
    vals = rng.uniform(-50, 400, (2, 63))
    open(os.path.join(g, 'joint.txt'), 'w').write('2\n' + '\n'.join(' '.join('%.6f' % v for v in row) for row in vals) + '\n')
    ds = datasets.MsraDataset('testing', 3, root)
    ann = ds.loadAnnotation()
    assert [a.name for a in ann] == ['1/000000_depth', '1/000001_depth']
    p = np.asarray(ann[1].pose)
    np.testing.assert_allclose(p[0::3], vals[1, 0::3], atol=1e-6)
    np.testing.assert_allclose(p[1::3], -vals[1, 1::3], atol=1e-6)
    np.testing.assert_allclose(p[2::3], -vals[1, 2::3], atol=1e-6)
    dm0 = datasets.read_msra_bin(os.path.join(g, '000000_depth.bin'))
    expect = np.zeros((cfg.h, cfg.w), np.float32); expect[top:bottom, left:right] = full[top:bottom, left:right]
    np.testing.assert_array_equal(dm0, expect)
    ds.cvtBin2Png()                                                                     # the empty second frame repeats the first
    for i in range(2):
        info, s = png.decode_png(open(os.path.join(g, '%06d_depth.png' % i), 'rb').read())
        np.testing.assert_array_equal(oracle_io.depth_from_samples(s, 1, 16).reshape(cfg.h, cfg.w), expect.astype(np.uint16).astype(np.float32))
    assert os.path.basename(ds.write_TFRecord(1)[0]) == 'P3-0-of-1'
    tr = datasets.MsraDataset('training', 3, root).filenames
    assert len(tr) == 801 and not any('/P3-' in f for f in tr) and len(ds.filenames) == 101


def test_nyu_adapter_joint_selection_and_boxes(tmp_path):
    import pickle
    import scipy.io as sio
    rng = np.random.default_rng(6)
    root = str(tmp_path)
    d = os.path.join(root, 'dataset', 'test')
    os.makedirs(d)
    cfg = datasets.NyuDataset.cfg
    joints = rng.uniform(-200, 900, (3, 2, 36, 3))
    sio.savemat(os.path.join(d, 'joint_data.mat'), {'joint_xyz': joints})
    bbx = rng.uniform(0, 400, (2, 5, 1)).astype(np.float32)
    pickle.dump([b for b in bbx], open(os.path.join(root, 'bbx.pkl'), 'wb'), protocol=2)
    frames = []
    for i in range(2):
        d16 = _depth_frame(rng, cfg.h, cfg.w, base=800)
        frames.append(d16)
        open(os.path.join(d, 'depth_1_%07d.png' % (i + 1)), 'wb').write(_pil_png(_nyu_rgb(d16, rng)))
    ds = datasets.NyuDataset('testing', root, bbx_path=os.path.join(root, 'bbx.pkl'))
    assert ds.jnt_num == 14 and ds.pose_dim == 42 and len(ds.filenames) == 17
    ann = ds.loadAnnotation()
    assert [a.name for a in ann] == ['depth_1_0000001.png', 'depth_1_0000002.png'] and len(ann[0].pose) == 108
    np.testing.assert_allclose(np.asarray(ann[1].pose).reshape(36, 3)[:, 1], -joints[0, 1, :, 1])
    paths = ds.write_TFRecord(1)
    recs = list(ds.records(False, epochs=1, files=paths))
    info, samples, pose, name, b = ds.parse_example(recs[1])
    keep = [0, 3, 6, 9, 12, 15, 18, 21, 24, 25, 27, 30, 31, 32]
    expect = joints[0, 1].copy(); expect[:, 1] *= -1
    np.testing.assert_allclose(pose.reshape(14, 3), expect[keep].astype(np.float32))
    np.testing.assert_array_equal(b, bbx[1].reshape(-1))
    np.testing.assert_array_equal(oracle_io.depth_from_samples(samples, 3, 8).reshape(480, 640), frames[1].astype(np.float32))


@pytest.mark.gpu
def test_dataset_batches_end_where_the_network_begins(gpu, tmp_path):
    """TFRecord shards -> batches(): decoded on the host, unpacked and cropped on the device -- the same crops, cameras
    and centres of mass as the oracle front-end applied to the Pillow-decoded frames; then through the network."""
    import torch
    from oracle import frontend as ofe
    rng = np.random.default_rng(8)
    root = str(tmp_path)
    frames = _make_icvl(root, rng, n=5)
    ds = datasets.IcvlDataset('testing', root)
    ds.loadAnnotation()
    paths = ds.write_TFRecord(2)
    got = list(ds.batches(3, torch.device('cuda', 0), shuffle=False, epochs=1, files=paths))
    assert [g[0].shape[0] for g in got] == [3, 2]
    k = 0
    cfg = np.asarray(ds.cfg, np.float32)
    for crops, poses, new_cfgs, coms, names in got:
        assert crops.shape[1:] == (128, 128, 1) and crops.is_cuda
        for b in range(crops.shape[0]):
            pose = np.asarray(ds.annotations[k].pose, np.float32)
            c_ref, _, cfg_ref = ofe.crop_from_xyz_pose(frames[k].astype(np.float32), pose, cfg, 128, 128, dataset='icvl')
            com_ref = ofe.center_of_mass(c_ref, cfg_ref)
            assert names[b] == ds.annotations[k].name
            np.testing.assert_allclose(crops[b, :, :, 0].cpu().numpy(), c_ref.reshape(128, 128), atol=2e-3)
            np.testing.assert_allclose(new_cfgs[b].cpu().numpy(), cfg_ref, rtol=1e-5, atol=1e-4)
            np.testing.assert_allclose(coms[b].cpu().numpy(), com_ref, rtol=1e-4, atol=1e-2)
            k += 1
    assert k == 5
    # the drivers' interface: a single pass for the test subset, then StopIteration
    ds.files_override = paths
    b0 = ds.batch(4, 0)
    b1 = ds.batch(4, 1)
    assert b0[0].shape[0] == 4 and b1[0].shape[0] == 1
    with pytest.raises(StopIteration):
        ds.batch(4, 2)


@pytest.mark.gpu
def test_test_driver_on_tfrecord_shards(gpu, tmp_path, monkeypatch):
    """``--data_dir``: the test driver (test_model.py:14-94) reads the reference's shard layout (testing-i-of-4 for ICVL)
    and writes one result line per frame, names with backslashes, ground truth = the labels of the shards."""
    from densereg_amd import flags
    from densereg_amd.model import hourglass_um_crop_tiny as M
    from densereg_amd.network import um_v1
    rng = np.random.default_rng(9)
    root = str(tmp_path / 'icvl')
    os.makedirs(root)
    _make_icvl(root, rng, n=6)
    w = datasets.IcvlDataset('testing', root)
    w.loadAnnotation()
    w.write_TFRecord(num_shards=4, num_threads=2)
    monkeypatch.chdir(tmp_path)
    flags.parse(['--dataset', 'icvl', '--num_stack', '1', '--fea_num', '64', '--is_train', 'False', '--batch_size', '4',
                 '--num_frames', '6', '--data_dir', root])
    try:
        tr, te = datasets.get_dataset('icvl', 'training', root), datasets.get_dataset('icvl', 'testing', root)
        eng = um_v1.get_engine(16, 128, 4, 0, False)
        eng.load_params(M._random_params(eng))
        model, (max_err, mean_err), out = M.run_test(tr, te)
        lines = open(out).read().splitlines()
        assert [l.split('\t')[0] for l in lines] == ['test_seq_1\\image_%04d.png' % i for i in range(6)]
        assert all(len(l.split('\t')) == 49 for l in lines) and len(mean_err) == 6 and np.isfinite(mean_err).all()
    finally:
        flags.parse([])
