#!/usr/bin/env python
"""One-off conv micro-benchmarks: python tools/conv_probe.py HW:Cin:Cout:k[:tile[:abl]] ...   (B=40, 30 launches each).
tile: -1 heuristic, 0 128x128, 1 64x128, 2 128x64, 3 64x64, 4 128x32, 5 64x64 BK64, 7 64x96, 8 64x160; abl: conv_igemm.h ABL."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densereg_amd import _lib  # noqa: E402


def main():
    lib = _lib.load()
    B = int(os.environ.get('PROBE_B', '40'))
    print('| HxW | Cin | Cout | k | tile | abl | us | TFLOP/s |')
    print('|---:|---:|---:|---:|---:|---:|---:|---:|')
    for spec in sys.argv[1:]:
        f = [int(v) for v in spec.split(':')]
        hw, cin, cout, k = f[:4]
        tile = f[4] if len(f) > 4 else -1
        abl = f[5] if len(f) > 5 else 0
        ms = C.c_float()
        rc = lib.dr_dbg_conv_bench(B, hw, hw, cin, cout, k, tile, abl, 30, C.byref(ms))
        flops = 2.0 * B * hw * hw * k * k * cin * cout
        print('| %d | %d | %d | %d | %d | %d | %s | %s |' % (hw, cin, cout, k, tile, abl, '%.1f' % (ms.value * 1e3) if rc == 0 else 'rc=%d' % rc,
                                                          '%.1f' % (flops / (ms.value * 1e-3) / 1e12) if rc == 0 else ''))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
