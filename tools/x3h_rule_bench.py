#!/usr/bin/env python
"""Where the 3x3 halo kernel (conv_x3h.h) beats the fp32-MFMA tiles: time per launch by shape and crops per launch.

    python tools/x3h_rule_bench.py
"""
import ctypes as C
import os
import sys

import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densereg_amd import _lib  # noqa: E402

SHAPES = [(32, 256, 256), (32, 128, 128), (32, 78, 78), (32, 65, 65), (32, 64, 64), (16, 64, 64), (16, 128, 128), (8, 64, 64), (8, 128, 128), (64, 16, 16), (32, 32, 32)]


def main():
    dbg = _lib.load_debug()
    ms = C.c_float()
    print('| 3x3 shape (HxW, Cin->Cout) | crops | row blocks x column blocks | fp32 us | x3 (halo) us | fp32 / x3 |')
    print('|---|---:|---:|---:|---:|---:|')
    for B in (200, 40, 8):
        for hw, cin, cout in SHAPES:
            row = []
            for mode in (0, 2):
                dbg.dr_dbg_force_x3(mode)
                rc = dbg.dr_dbg_conv_bench(B, hw, hw, cin, cout, 3, -1, 0, 10, C.byref(ms))
                assert rc == 0, rc
                row.append(ms.value * 1e3)
            dbg.dr_dbg_force_x3(-1)
            nb = -(-B * hw * hw // 128)
            print('| %dx%d %d->%d | %d | %d x %d | %.1f | %.1f | %.2fx |' % (hw, hw, cin, cout, B, nb, -(-cout // 128), row[0], row[1], row[0] / row[1]))
            sys.stdout.flush()


if __name__ == '__main__':
    main()
