#!/usr/bin/env python
"""Narrow-output conv layers (N = 65..96, 129..160): the 128x32 tile against the 64x96 / 64x160 K-split tiles, fp32 and bf16.

    python tools/conv_narrow_bench.py > gpurun_out/conv_narrow.md
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densereg_amd import _lib  # noqa: E402


def main():
    lib = _lib.load_debug()
    # (B, HxW, Cin, Cout, k): S=2 F=128 at B=40 and B=8, config 5 (S=4 F=256, 64x64 maps) at B=40
    shapes = [(40, 32, 78, 78, 3), (40, 32, 65, 65, 3), (40, 32, 156, 78, 1), (40, 32, 131, 65, 1), (40, 32, 256, 156, 1), (40, 32, 78, 156, 1),
              (40, 32, 85, 85, 3), (8, 32, 78, 78, 3), (8, 32, 156, 78, 1), (4, 32, 78, 78, 3),
              (40, 64, 129, 129, 3), (40, 64, 259, 129, 1), (40, 64, 142, 142, 3), (40, 64, 284, 142, 1), (40, 64, 256, 142, 1)]
    print('| B | HxW | Cin | Cout | k | 128x32 us | narrow us | speed-up | narrow TFLOP/s | bf16: 128x32 us | narrow us |')
    print('|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|')
    for B, hw, cin, cout, k in shapes:
        np_ = -(-cout // 32) * 32
        narrow = 7 if np_ == 96 else 8
        res = []
        for bf in (0, 1):
            lib.dr_dbg_force_bf16(bf)
            for tile in (4, narrow):
                ms = C.c_float()
                rc = lib.dr_dbg_conv_bench(B, hw, hw, cin, cout, k, tile, 0, 30, C.byref(ms))
                res.append(ms.value * 1e3 if rc == 0 else float('nan'))
        lib.dr_dbg_force_bf16(0)
        flops = 2.0 * B * hw * hw * k * k * cin * cout
        print('| %d | %d | %d | %d | %d | %.1f | %.1f | %.2fx | %.1f | %.1f | %.1f |' % (B, hw, cin, cout, k, res[0], res[1], res[0] / res[1],
                                                                                   flops / res[1] / 1e6, res[2], res[3]))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
