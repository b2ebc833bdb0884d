#!/usr/bin/env python
"""Micro-benchmark of the BatchReNorm streaming kernels (bytes moved / time) on an MI355X.

    python tools/bn_bench.py > gpurun_out/bn_bench.md
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densereg_amd import _lib  # noqa: E402


def main():
    lib = _lib.load_debug()
    print('| M | C | reduce grid | apply us | GB/s | bwd reduce us | GB/s | bwd apply us | GB/s |')
    print('|---:|---:|---:|---:|---:|---:|---:|---:|---:|')
    for M, Cc in ((40960, 256), (40960, 512), (40960, 128), (40960, 64), (10240, 256), (2560, 256), (640, 256), (160, 256)):
        for rb in (0, 128, 256, 1024, 2048):
            us = (C.c_float * 3)()
            rc = lib.dr_dbg_bn_bench(M, Cc, rb, 20, us)
            if rc:
                print('| %d | %d | %d | rc=%d |' % (M, Cc, rb, rc))
                continue
            t = M * Cc * 4.0
            print('| %d | %d | %s | %.1f | %.0f | %.1f | %.0f | %.1f | %.0f |' % (
                M, Cc, rb or 'auto', us[0], 2 * t / us[0] / 1e3, us[1], 2 * t / us[1] / 1e3, us[2], 3 * t / us[2] / 1e3))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
