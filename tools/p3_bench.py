#!/usr/bin/env python
"""conv_p3.h (x3 products on a P3-stored input) against conv_x3_kernel: time per launch, shape by shape.

    [DR_P3_VARIANT=n] python tools/p3_bench.py [B]        (B crops per launch at 32x32, default 200 = one accumulation window)
"""
import ctypes as C
import os
import sys

import torch  # noqa: F401  (before the library: torch brings its own HIP runtime, which must be the first one loaded)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densereg_amd import _lib  # noqa: E402

SHAPES = [(32, 256, 256, 3), (32, 128, 128, 3), (32, 512, 512, 1), (32, 512, 256, 1), (32, 256, 512, 1), (32, 256, 128, 1), (32, 128, 256, 1),
          (32, 128, 128, 1), (32, 515, 512, 1), (64, 256, 256, 3)]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dbg = _lib.load_debug()
    ms = C.c_float()
    print('| shape (HxW, Cin->Cout, k) at %d crops, DR_P3_VARIANT=%s | x3 us | TFLOP/s | p3 us | TFLOP/s | p3 / x3 |' % (B, os.environ.get('DR_P3_VARIANT', '0')))
    print('|---|---:|---:|---:|---:|---:|')
    for hw, cin, cout, k in SHAPES:
        b = B if hw == 32 else max(1, B // 4)
        fl = 2.0 * b * hw * hw * k * k * cin * cout
        row = []
        for mode in (2, 6):
            dbg.dr_dbg_force_x3(mode)
            rc = dbg.dr_dbg_conv_bench(b, hw, hw, cin, cout, k, -1, 0, 10, C.byref(ms))
            assert rc == 0, rc
            row.append(ms.value * 1e3)
        dbg.dr_dbg_force_x3(-1)
        print('| %dx%d %d->%d k%d | %.1f | %.1f | %.1f | %.1f | %.2fx |' % (hw, hw, cin, cout, k, row[0], fl / row[0] / 1e6, row[1], fl / row[1] / 1e6, row[0] / row[1]))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
