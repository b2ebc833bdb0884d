#!/usr/bin/env python
"""Run-to-run stability of the two-slot pipeline: the trajectory of tests/test_pipeline.py N times at depth 1 and 2, checksums of
losses / moving statistics / parameters per run.  python tests/stress_pipeline.py [runs]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.common import GpuBackend  # noqa: E402
from tests.test_pipeline import _case, _trajectory  # noqa: E402


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    be = GpuBackend()
    cfg, params, batches, B = _case(be)
    ref = None
    for depth in (1, 2, 2, 2, 1, 2, 2, 2)[:runs]:
        lo, p, _ = _trajectory(be, cfg, params, batches, B, depth)
        mm = {k: v for k, v in p.items() if k.endswith('moving_mean')}
        w = {k: v for k, v in p.items() if k.endswith('weights')}
        cs_mm = sum(float(np.abs(v).sum(dtype=np.float64)) for v in mm.values())
        cs_w = sum(float(np.abs(v).sum(dtype=np.float64)) for v in w.values())
        if ref is None:
            ref = p
        worst = max(mm, key=lambda k: float(np.abs(mm[k] - ref[k]).max() / (np.abs(ref[k]).max() + 1e-9)))
        print('depth %d: losses %s | sum|moving_mean| %.6f sum|w| %.6f | worst moving_mean vs first run: %s %.3e' % (
            depth, ' '.join('%.4f' % float(l[0]) for l in lo), cs_mm, cs_w, worst,
            float(np.abs(mm[worst] - ref[worst]).max() / (np.abs(ref[worst]).max() + 1e-9))))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
