#!/usr/bin/env python
"""conv_x3_kernel on the wide 1x1 layers under the current DR_X3_BIG (0 = 128-column blocks, 1 = 256 columns as eight waves of 64x64,
2 = 256 columns as sixteen waves of 64x32): time per launch, shape by shape.  One process per setting (the switch is read once).

    DR_X3_BIG=1 python tools/x3_bn256_bench.py [B]      (B crops per launch at 32x32, default 200 = one accumulation window)
"""
import ctypes as C
import os
import sys

import torch  # noqa: F401  (before the library: torch brings its own HIP runtime, which must be the first one loaded)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densereg_amd import _lib  # noqa: E402

SHAPES = [(32, 512, 512, 1), (32, 515, 512, 1), (32, 256, 512, 1), (32, 512, 256, 1), (32, 128, 256, 1), (32, 256, 128, 1), (32, 128, 128, 1), (32, 156, 256, 1), (32, 78, 256, 1),
          (64, 256, 256, 3), (32, 256, 256, 3)]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dbg = _lib.load_debug()
    ms = C.c_float()
    print('| DR_X3_BIG=%s, %d crops | us | TFLOP/s |' % (os.environ.get('DR_X3_BIG', '0'), B))
    print('|---|---:|---:|')
    dbg.dr_dbg_force_x3(2)
    for hw, cin, cout, k in SHAPES:
        b = B if hw == 32 else max(1, B // 4)
        fl = 2.0 * b * hw * hw * k * k * cin * cout
        best = 1e9
        for _ in range(3):
            rc = dbg.dr_dbg_conv_bench(b, hw, hw, cin, cout, k, -1, 0, 10, C.byref(ms))
            assert rc == 0, rc
            best = min(best, ms.value * 1e3)
        print('| %dx%d %d->%d k%d | %.1f | %.1f |' % (hw, hw, cin, cout, k, best, fl / best / 1e6))
        sys.stdout.flush()
    dbg.dr_dbg_force_x3(-1)


if __name__ == '__main__':
    main()
