#!/usr/bin/env python
"""Micro-benchmark of the bf16 matrix-core conv kernels (conv_igemm.h, BF = 1) next to the fp32 ones, per tile, on an MI355X.

    python tools/conv_bench_bf16.py > gpurun_out/conv_bench_bf16.md
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densereg_amd import _lib  # noqa: E402

TILES = {-1: 'auto', 0: '128x128', 1: '64x128', 2: '128x64', 3: '64x64', 4: '128x32'}


def main():
    lib = _lib.load_debug()
    B = 40
    shapes = [(32, 256, 256, 3), (32, 128, 128, 3), (32, 512, 512, 1), (32, 515, 512, 1), (32, 512, 256, 1), (32, 256, 512, 1),
              (32, 256, 128, 1), (32, 128, 64, 1), (32, 64, 64, 3), (32, 80, 80, 3), (16, 64, 64, 3), (8, 64, 64, 3), (64, 16, 16, 3)]
    print('| HxW | Cin | Cout | k | tile | fp32 us | bf16 us | speed-up | bf16 TFLOP/s | algorithmic GB/s (fp32 in + out) |')
    print('|---:|---:|---:|---:|---|---:|---:|---:|---:|---:|')
    for hw, cin, cout, k in shapes:
        flops = 2.0 * B * hw * hw * k * k * cin * cout
        nbytes = 4.0 * B * hw * hw * (cin + cout)
        np_ = -(-cout // 32) * 32
        tiles = [-1] + ([0, 1, 2, 3] if np_ % 128 == 0 else [2, 3] if np_ % 64 == 0 else [4])
        for tile in tiles:
            t = []
            for bf in (0, 1):
                lib.dr_dbg_force_bf16(bf)
                ms = C.c_float()
                rc = lib.dr_dbg_conv_bench(B, hw, hw, cin, cout, k, tile, 0, 20, C.byref(ms))
                t.append(ms.value * 1e3 if rc == 0 else float('nan'))
            lib.dr_dbg_force_bf16(0)
            print('| %d | %d | %d | %d | %s | %.1f | %.1f | %.2f | %.0f | %.0f |' % (
                hw, cin, cout, k, TILES[tile], t[0], t[1], t[0] / t[1], flops / (t[1] * 1e-6) / 1e12, nbytes / (t[1] * 1e-6) / 1e9))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
