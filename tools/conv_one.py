#!/usr/bin/env python
"""Launch one conv shape a few times (for rocprofv3 --pmc passes on a single kernel).

    python tools/conv_one.py HW CIN COUT K [TILE] [ITERS]          (PROBE_B: crops per launch, default 40; PROBE_X3: dr_dbg_force_x3 mode)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densereg_amd import _lib  # noqa: E402

hw, cin, cout, k = (int(v) for v in sys.argv[1:5])
tile = int(sys.argv[5]) if len(sys.argv) > 5 else -1
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 3
ms = C.c_float()
dbg = _lib.load_debug()
if os.environ.get('PROBE_X3'):
    assert dbg.dr_dbg_force_x3(int(os.environ['PROBE_X3'])) == 0
rc = dbg.dr_dbg_conv_bench(int(os.environ.get('PROBE_B', '40')), hw, hw, cin, cout, k, tile, 0, iters, C.byref(ms))
print('rc', rc, 'us', ms.value * 1e3)
