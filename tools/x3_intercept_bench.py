#!/usr/bin/env python
"""conv_x3_kernel (1x1, 204 800 rows): time against K for N = 128 / 256 / 512 -- the slope is the K loop, the intercept what a launch
costs beyond it (prologue, epilogue, output stores).  DR_X3_ABL=3 (debug library): the same without the epilogue's stores.

    python tools/x3_intercept_bench.py [B]
"""
import ctypes as C
import os
import sys

import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densereg_amd import _lib  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dbg = _lib.load_debug()
    ms = C.c_float()
    print('| DR_X3_ABL=%s, %d crops at 32x32: us per launch | K=16 | 64 | 128 | 256 | 512 | slope us / K-tile | intercept us | output MB |' % (os.environ.get('DR_X3_ABL', '0'), B))
    print('|---|---:|---:|---:|---:|---:|---:|---:|---:|')
    dbg.dr_dbg_force_x3(2)
    for n in (128, 256, 512):
        ts = []
        for kk in (16, 64, 128, 256, 512):
            best = 1e9
            for _ in range(3):
                rc = dbg.dr_dbg_conv_bench(B, 32, 32, kk, n, 1, -1, 0, 10, C.byref(ms))
                assert rc == 0, rc
                best = min(best, ms.value * 1e3)
            ts.append(best)
        slope = (ts[4] - ts[2]) / 24.0
        print('| N=%d | %s | %.2f | %.1f | %.0f |' % (n, ' | '.join('%.1f' % t for t in ts), slope, ts[2] - 8 * slope, B * 1024 * n * 4 / 1e6))
        sys.stdout.flush()
    dbg.dr_dbg_force_x3(-1)


if __name__ == '__main__':
    main()
