#!/usr/bin/env python
"""conv_wgrad_x3.h against the fp32-MFMA weight-gradient kernels: kernel + slab fold, the planner's slab count, per shape.

    python tools/wgrad_x3_bench.py [B]          (crops per launch at the given map size, default 200)
"""
import ctypes as C
import os
import sys

import torch  # noqa: F401  (first: its HIP runtime must be the one loaded)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densereg_amd import _lib  # noqa: E402

SHAPES = [(32, 256, 256, 3, 128), (32, 128, 128, 3, 128), (32, 512, 512, 1, 128), (32, 512, 256, 1, 128), (32, 256, 512, 1, 128), (32, 515, 512, 1, 64),
          (32, 128, 256, 1, 128), (32, 256, 128, 1, 128), (32, 128, 128, 1, 128), (32, 64, 64, 3, 64), (32, 128, 64, 1, 64), (32, 64, 128, 1, 64),
          (16, 64, 64, 3, 64)]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    lib = _lib.load_debug()
    print('| HxW, Cin->Cout, k, tile at %d crops | slabs | fp32 MFMA us | TFLOP/s | x3 us | TFLOP/s | x3 / fp32 |' % B)
    print('|---|---:|---:|---:|---:|---:|---:|')
    for hw, cin, cout, k, T in SHAPES:
        fl = 2.0 * B * hw * hw * k * k * cin * cout
        row, used = [], C.c_int()
        for mode in (0, 2):
            lib.dr_dbg_force_x3(mode)
            us = C.c_float()
            rc = lib.dr_dbg_wgrad_bench(B, hw, hw, cin, cout, k, T, 0, 10, C.byref(us), C.byref(used))
            assert rc == 0, rc
            row.append(us.value)
        lib.dr_dbg_force_x3(-1)
        print('| %dx%d %d->%d k%d T%d | %d | %.1f | %.1f | %.1f | %.1f | %.2fx |' % (hw, hw, cin, cout, k, T, used.value, row[0], fl / row[0] / 1e6,
                                                                                  row[1], fl / row[1] / 1e6, row[0] / row[1]))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
