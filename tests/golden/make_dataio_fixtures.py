#!/usr/bin/env python
"""Generate tests/golden/dataio_fixtures.npz: depth frames as PNG streams written by an INDEPENDENT codec (Pillow) with
the sample arrays they must decode to, and one TFRecord shard written by densereg_amd.data.tfrecord with its contents.

    python tests/golden/make_dataio_fixtures.py

The PNG bytes pin ``densereg_amd.data.png`` / ``dr_png_unfilter`` / ``oracle.dataio`` against Pillow's encoder (adaptive
filter choice, so several filter types occur in each stream); the record bytes pin the framing and the Example codec
against what this repository wrote when the fixture was made (format parity with TensorFlow itself is unpinned: no
TF-written file exists in this environment).
"""
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from PIL import Image
    from densereg_amd.data import tfrecord
    rng = np.random.default_rng(2024)
    yy, xx = np.mgrid[0:24, 0:40]
    d16 = np.clip(600 + 90 * np.sin(xx / 5.0) * np.cos(yy / 3.0) + rng.normal(0, 4, (24, 40)), 0, 65535).astype(np.uint16)
    d16[(yy - 12) ** 2 + (xx - 20) ** 2 > 120] = 0
    rgb = np.zeros((24, 40, 3), np.uint8)
    rgb[..., 0] = rng.integers(0, 256, (24, 40))
    rgb[..., 1] = d16 >> 8
    rgb[..., 2] = d16 & 0xFF

    def png(a, **kw):
        buf = io.BytesIO()
        Image.fromarray(a).save(buf, format='PNG', **kw)
        return np.frombuffer(buf.getvalue(), np.uint8)

    pose = rng.uniform(-100, 500, 48).astype(np.float32)
    rec = tfrecord.make_example({'name': b'test_seq_1/image_0000.png', 'xyz_pose': pose, 'png16': png(d16).tobytes()})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_tmp_shard')
    tfrecord.write_records(path, [rec, b'second'])
    shard = np.frombuffer(open(path, 'rb').read(), np.uint8)
    os.remove(path)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dataio_fixtures.npz'),
                        depth16=d16, png_grey16=png(d16), png_grey16_opt=png(d16, optimize=True), rgb=rgb, png_rgb8=png(rgb),
                        pose=pose, shard=shard)
    print('wrote dataio_fixtures.npz: grey16 %d B, rgb8 %d B, shard %d B' % (png(d16).size, png(rgb).size, shard.size))


if __name__ == '__main__':
    main()
