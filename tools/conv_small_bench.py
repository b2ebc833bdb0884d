#!/usr/bin/env python
"""Micro-benchmark of the conv kernels on the layers below 32x32 pixels (grids that cannot fill the chip): the 64x64
tile, its fat K-tile variant and the split-K 32x32 kernel, fp32 and bf16 matrix cores, B = 40, on an MI355X.

    python tools/conv_small_bench.py > gpurun_out/conv_small_bench.md
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densereg_amd import _lib  # noqa: E402

TILES = {-1: 'auto', 3: '64x64', 5: '64x64 BK64', 6: 'split-K 32x32', 4: '128x32'}


def main():
    lib = _lib.load_debug()
    B = 40
    shapes = [(hw, cin, cout, k) for hw in (16, 8, 4, 2) for (cin, cout, k) in ((64, 64, 3), (128, 64, 1), (64, 128, 1), (128, 128, 1))]
    shapes += [(32, 64, 64, 3), (32, 128, 64, 1), (32, 64, 128, 1), (64, 16, 16, 3), (64, 32, 16, 1)]
    print('| HxW | Cin | Cout | k | variant | fp32 us | bf16 us |')
    print('|---:|---:|---:|---:|---|---:|---:|')
    for hw, cin, cout, k in shapes:
        np_ = -(-cout // 32) * 32
        tiles = [-1, 6] + ([3, 5] if np_ % 64 == 0 else [4])
        for tile in tiles:
            t = []
            for bf in (0, 1):
                if bf and tile == 5:
                    t.append(float('nan'))
                    continue
                lib.dr_dbg_force_bf16(bf)
                ms = C.c_float()
                rc = lib.dr_dbg_conv_bench(B, hw, hw, cin, cout, k, tile, 0, 30, C.byref(ms))
                t.append(ms.value * 1e3 if rc == 0 else float('nan'))
            lib.dr_dbg_force_bf16(0)
            print('| %d | %d | %d | %d | %s | %.1f | %.1f |' % (hw, cin, cout, k, TILES[tile], t[0], t[1]))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
