#!/usr/bin/env python
"""Weight-gradient kernel + slab fold by slab count (calibrates train_exec.inc::wgrad_plan) on an MI355X.

    python tools/wgrad_bench.py > gpurun_out/wgrad_bench.md
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densereg_amd import _lib  # noqa: E402


def main():
    lib = _lib.load_debug()
    B = int(os.environ.get('PROBE_B', '40'))          # crops per launch (200 = a five-micro-batch window)
    shapes = [(32, 78, 78, 3), (32, 65, 65, 3), (32, 96, 96, 3), (32, 256, 256, 3), (32, 128, 128, 3), (32, 512, 512, 1), (32, 128, 256, 1),
              (32, 78, 256, 1), (32, 128, 128, 1), (32, 128, 64, 1), (16, 64, 64, 3), (16, 128, 64, 1), (8, 64, 64, 3)]
    if os.environ.get('PROBE_SHAPES'):                   # "hw:cin:cout:k,..." instead of the list above
        shapes = [tuple(int(v) for v in sp.split(':')) for sp in os.environ['PROBE_SHAPES'].split(',')]
    slabs = [int(v) for v in os.environ.get('PROBE_NS', '0,4,8,16,32,64,128,256').split(',')]
    tiles = [int(v) for v in os.environ.get('PROBE_T', '128,64,96').split(',')]
    print('| HxW | Cin | Cout | k | T | slabs | us (kernel + fold) | TFLOP/s |')
    print('|---:|---:|---:|---:|---:|---:|---:|---:|')
    for hw, cin, cout, k in shapes:
        flops = 2.0 * B * hw * hw * k * k * cin * cout
        for T in tiles:
            for ns in slabs:
                us, used = C.c_float(), C.c_int()
                rc = lib.dr_dbg_wgrad_bench(B, hw, hw, cin, cout, k, T, ns, 10, C.byref(us), C.byref(used))
                if rc:
                    continue
                print('| %d | %d | %d | %d | %d | %s%d | %.1f | %.1f |' % (hw, cin, cout, k, T, 'plan ' if ns == 0 else '', used.value,
                                                                       us.value, flops / us.value / 1e6))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
