#!/usr/bin/env python
"""Micro-benchmark of the conv kernel on the shapes that carry the training step, heuristic tile, current environment
(run once per setting of DR_CONV_GLDS etc.):   DR_CONV_GLDS=1 python tools/conv_ab.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densereg_amd import _lib  # noqa: E402


def main():
    lib = _lib.load_debug()
    B = 40
    shapes = [(32, 256, 256, 3), (32, 512, 512, 1), (32, 515, 512, 1), (32, 128, 128, 3), (32, 512, 256, 1), (32, 256, 512, 1),
              (32, 256, 128, 1), (32, 128, 256, 1), (32, 128, 128, 1), (32, 64, 64, 3), (32, 128, 64, 1), (32, 64, 128, 1),
              (32, 160, 256, 1), (32, 80, 80, 3), (16, 64, 64, 3)]
    tag = ' '.join('%s=%s' % (k, os.environ[k]) for k in ('DR_CONV_GLDS',) if k in os.environ) or 'default'
    print('| %s | HxW | Cin | Cout | k | us | TFLOP/s |' % tag)
    for hw, cin, cout, k in shapes:
        ms = C.c_float()
        rc = lib.dr_dbg_conv_bench(B, hw, hw, cin, cout, k, -1, 0, 30, C.byref(ms))
        flops = 2.0 * B * hw * hw * k * k * cin * cout
        print('| | %d | %d | %d | %d | %s | %s |' % (hw, cin, cout, k, '%.1f' % (ms.value * 1e3) if rc == 0 else 'rc=%d' % rc,
                                                     '%.1f' % (flops / (ms.value * 1e-3) / 1e12) if rc == 0 else ''))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
