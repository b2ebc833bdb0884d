#!/usr/bin/env python
"""Per-shape micro-benchmark of the implicit-GEMM conv kernel (tile shapes + ablations) on an MI355X.

    python tools/conv_bench.py > gpurun_out/conv_bench.md
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densereg_amd import _lib  # noqa: E402

TILES = {-1: 'auto', 0: '128x128', 1: '64x128', 2: '128x64', 3: '64x64', 4: '128x32', 5: '64x64 BK64'}
ABL = {0: '', 1: 'no-refill', 2: 'no-mfma', 3: 'no-store', 5: '+residual', 6: 'zero operands', 7: '64x128 no-refill', 8: 'loads, no LDS writes', 9: 'LDS writes, no loads'}


def main():
    lib = _lib.load_debug()
    B = 40
    shapes = [(32, 256, 256, 3), (32, 128, 128, 3), (32, 512, 512, 1), (32, 515, 512, 1), (32, 512, 256, 1),
              (32, 256, 512, 1), (32, 256, 128, 1), (32, 128, 256, 1), (32, 128, 64, 1), (32, 64, 128, 1),
              (32, 64, 64, 3), (32, 160, 256, 1), (32, 80, 80, 3), (32, 65, 65, 3), (32, 131, 65, 1), (32, 160, 80, 1), (16, 64, 64, 3), (16, 128, 64, 1), (16, 64, 128, 1), (8, 64, 64, 3),
              (8, 128, 64, 1), (4, 64, 64, 3), (2, 64, 64, 3), (2, 128, 64, 1), (64, 32, 64, 1), (64, 16, 16, 3)]
    print('sustained fp32 MFMA rate, register-only chains (nominal peak 157.3 TFLOP/s):\n')
    print('| waves/SIMD | data | TFLOP/s |\n|---:|---|---:|')
    for wps in (1, 2, 4):
        for zero in (0, 1):
            tf = C.c_float()
            rc = lib.dr_dbg_mfma_peak(20000, wps, zero, C.byref(tf))
            print('| %d | %s | %s |' % (wps, 'zeros' if zero else 'varied', '%.1f' % tf.value if rc == 0 else 'rc=%d' % rc))
    print()
    print('| HxW | Cin | Cout | k | variant | us | TFLOP/s | % of 157.3 |')
    print('|---:|---:|---:|---:|---|---:|---:|---:|')
    for hw, cin, cout, k in shapes:
        flops = 2.0 * B * hw * hw * k * k * cin * cout
        np_ = -(-cout // 32) * 32
        variants = [(-1, 0), (-1, 5)]
        if np_ % 128 == 0:
            variants += [(0, 0), (1, 0), (2, 0), (3, 0), (0, 1), (0, 2), (1, 6), (1, 7)]
        elif np_ % 64 == 0:
            variants += [(2, 0), (3, 0), (5, 0)]
        for tile, abl in variants:
            ms = C.c_float()
            rc = lib.dr_dbg_conv_bench(B, hw, hw, cin, cout, k, tile, abl, 20, C.byref(ms))
            if rc != 0:
                print('| %d | %d | %d | %d | %s %s | rc=%d | | |' % (hw, cin, cout, k, TILES[tile], ABL[abl], rc))
                continue
            tf = flops / (ms.value * 1e-3) / 1e12
            print('| %d | %d | %d | %d | %s %s | %.1f | %.1f | %.0f |' % (hw, cin, cout, k, TILES[tile], ABL[abl],
                                                                      ms.value * 1e3, tf, 100 * tf / 157.3))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
