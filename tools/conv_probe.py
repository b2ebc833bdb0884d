#!/usr/bin/env python
"""One-off conv micro-benchmarks: python tools/conv_probe.py HW:Cin:Cout:k[:tile[:abl]] ...   (B=40, 30 launches each).
tile: -1 heuristic, 0 128x128, 1 64x128, 2 128x64, 3 64x64, 4 128x32, 5 64x64 BK64, 6 split-K 32x32, 7 64x96, 8 64x160.
abl (dr_dbg_conv_bench; 0 = product kernel of the given tile): 1-3 the 128x128 tile without refill / with VALU instead of MFMA /
without epilogue stores; 5 product kernel + residual add; 6 all-zero operands; on the 64x128 tile: 7 no refill, 8 refill loads
without LDS writes, 9 LDS writes without loads, 10 = 7 without the barrier, 11 = 7 without fragment reads, 12 = neither (the bare
MFMA loop), 13 = 12 without epilogue stores."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densereg_amd import _lib  # noqa: E402


def main():
    lib = _lib.load_debug()
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
