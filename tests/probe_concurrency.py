#!/usr/bin/env python
"""Is a depth-1 engine's trajectory bit-stable while an UNRELATED stream keeps the GPU busy (torch.matmul on its own tensors)?
If not, the hazard is inside a single stream's kernel chain under concurrency, not in the two-slot bookkeeping."""
import os
import sys
import threading

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.common import GpuBackend  # noqa: E402
from tests.test_pipeline import _case, _trajectory  # noqa: E402


def main():
    be = GpuBackend()
    cfg, params, batches, B = _case(be)
    dev = be.device
    noise = sys.argv[1] if len(sys.argv) > 1 else 'matmul'
    stop = False
    side = torch.cuda.Stream(dev)
    a = torch.randn(4096, 4096, device=dev)
    b = torch.randn(4096, 4096, device=dev)
    big = torch.empty(64 * 1024 * 1024, device=dev)

    def hammer():
        torch.cuda.set_device(dev)
        with torch.cuda.stream(side):
            while not stop:
                for _ in range(20):
                    if noise == 'matmul':
                        torch.matmul(a, b)
                    else:
                        big.mul_(1.0001)
                side.synchronize()
    ref = None
    for run, busy in enumerate((False, True, True, True, False, True)):
        th = None
        stop = False
        if busy:
            th = threading.Thread(target=hammer)
            th.start()
        lo, p, _ = _trajectory(be, cfg, params, batches, B, 1)
        stop = True
        if th:
            th.join()
        cs = sum(float(np.abs(v).sum(dtype=np.float64)) for k, v in p.items() if k.endswith('moving_mean'))
        print('run %d noise=%s: losses %s | sum|moving_mean| %.6f' % (run, noise if busy else 'none', ' '.join('%.4f' % float(l[0]) for l in lo), cs))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
