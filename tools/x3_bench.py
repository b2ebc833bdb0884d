#!/usr/bin/env python
"""conv_x3.h against the fp32-MFMA kernels: time per launch and error against an fp64 reference, shape by shape.

    python tools/x3_bench.py [B]          (B crops per launch at 32x32, default 200 = one accumulation window)
"""
import ctypes as C
import os
import sys

import numpy as np
import torch  # noqa: F401  (before the library: torch brings its own HIP runtime, which must be the first one loaded)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densereg_amd import _lib  # noqa: E402

SHAPES = [(32, 256, 256, 3), (32, 128, 128, 3), (32, 512, 512, 1), (32, 512, 256, 1), (32, 256, 512, 1), (32, 256, 128, 1), (32, 128, 256, 1),
          (32, 128, 128, 1), (32, 515, 512, 1), (32, 78, 78, 3), (32, 65, 65, 3), (32, 156, 78, 1), (32, 131, 65, 1), (32, 64, 64, 3), (32, 128, 64, 1), (32, 64, 128, 1), (16, 64, 64, 3), (16, 128, 64, 1)]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dbg = _lib.load_debug()
    ms = C.c_float()
    print('| shape (HxW, Cin->Cout, k) at %d crops | fp32 MFMA us | TFLOP/s | x3 us | TFLOP/s | x3 four waves us | TFLOP/s | x3 ring us | TFLOP/s | p3 us | TFLOP/s | x3 / fp32 | p3 / x3 |' % B)
    print('|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|')
    for hw, cin, cout, k in SHAPES:
        fl = 2.0 * B * hw * hw * k * k * cin * cout
        row = []
        for mode in (0, 2, 5, 4, 6):
            dbg.dr_dbg_force_x3(mode)
            rc = dbg.dr_dbg_conv_bench(B, hw, hw, cin, cout, k, -1, 0, 10, C.byref(ms))
            assert rc == 0, rc
            row.append(ms.value * 1e3)
        dbg.dr_dbg_force_x3(-1)
        print('| %dx%d %d->%d k%d | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.2fx | %.2fx |' % (
            hw, hw, cin, cout, k, row[0], fl / row[0] / 1e6, row[1], fl / row[1] / 1e6, row[2], fl / row[2] / 1e6, row[3], fl / row[3] / 1e6,
            row[4], fl / row[4] / 1e6, row[0] / row[1], row[1] / row[4]))
        sys.stdout.flush()
    # errors against fp64 on one big layer (network-like operands: post-ReLU activations, He weights)
    from tests.common import GpuBackend, ref_conv2d
    be = GpuBackend()
    rng = np.random.default_rng(3)
    for (b, hw, cin, cout, k) in ((2, 32, 256, 256, 3), (4, 32, 512, 512, 1)):
        x = np.maximum(rng.standard_normal((b, hw, hw, cin)), 0).astype(np.float32)
        w = (rng.standard_normal((k, k, cin, cout)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
        yr, _ = ref_conv2d(x, w)
        out = []
        for mode in (0, 2, 5, 3):
            be.dbg.dr_dbg_force_x3(mode)
            y = be.conv2d(x, w)
            e = np.abs(y - yr)
            out.append('%s: max %.2e rms %.2e' % ({0: 'fp32', 2: 'x3', 5: 'x3 four waves', 3: 'x3 one-acc'}[mode], e.max() / np.abs(yr).max(), np.sqrt((e ** 2).mean()) / np.abs(yr).max()))
        be.dbg.dr_dbg_force_x3(-1)
        print('error vs fp64 (of the range), %dx%d %d->%d k%d B=%d: %s' % (hw, hw, cin, cout, k, b, ' | '.join(out)))


if __name__ == '__main__':
    main()
