#!/usr/bin/env python
"""conv_x3h.h (3x3 with the haloed tile resident in LDS) against conv_x3_kernel: time per launch at 200 crops.

    python tools/x3h_bench.py [B]
"""
import ctypes as C
import os
import sys

import torch  # noqa: F401  (before the library: torch brings its own HIP runtime, which must be the first one loaded)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densereg_amd import _lib  # noqa: E402

SHAPES = [(32, 256, 256, 3), (32, 128, 128, 3), (32, 78, 78, 3), (32, 65, 65, 3), (32, 64, 64, 3), (16, 256, 256, 3), (16, 64, 64, 3), (32, 64, 128, 3)]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dbg = _lib.load_debug()
    ms = C.c_float()
    print('| shape (HxW, Cin->Cout, k) at %d crops | x3 us | TFLOP/s | x3h us | TFLOP/s | p3 us | TFLOP/s | x3h / x3 |' % B)
    print('|---|---:|---:|---:|---:|---:|---:|---:|')
    for hw, cin, cout, k in SHAPES:
        fl = 2.0 * B * hw * hw * k * k * cin * cout
        row = []
        for mode in (7, 2, 6):
            dbg.dr_dbg_force_x3(mode)
            rc = dbg.dr_dbg_conv_bench(B, hw, hw, cin, cout, k, -1, 0, 10, C.byref(ms))
            assert rc == 0, rc
            row.append(ms.value * 1e3)
        dbg.dr_dbg_force_x3(-1)
        print('| %dx%d %d->%d k%d | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.2fx |' % (hw, hw, cin, cout, k, row[0], fl / row[0] / 1e6, row[1], fl / row[1] / 1e6,
                                                                                   row[2], fl / row[2] / 1e6, row[0] / row[1]))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
