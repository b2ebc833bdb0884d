#!/usr/bin/env python
"""Throughput of the input front-end kernels on an MI355X (torch events on the current stream; a call includes
the host mirror's three output allocations and the ctypes launch, i.e. what a data loader would pay).

    python tools/frontend_bench.py > gpurun_out/frontend_bench.md
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densereg_amd.data import preprocess as P  # noqa: E402


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters          # us


def main():
    rng = np.random.default_rng(0)
    print('| kernel | frames | geometry | us / call | frames / s | algorithmic GB/s |')
    print('|---|---:|---|---:|---:|---:|')
    for name, H, W, J, fx in (('icvl', 240, 320, 16, 241.42), ('nyu', 480, 640, 14, 588.03)):
        for B in (40, 320):
            dms = torch.from_numpy(rng.uniform(300, 900, (B, H, W)).astype(np.float32)).cuda()
            cfg = torch.tensor([fx, fx, W / 2, H / 2, W, H], dtype=torch.float32).repeat(B, 1).cuda()
            uv = rng.uniform(-60, 60, (B, J, 2)) + np.array([W / 2, H / 2])
            z = rng.uniform(350, 500, (B, J))
            pose = np.stack([(uv[..., 0] - W / 2) * z / fx, (uv[..., 1] - H / 2) * z / fx, z], -1).reshape(B, -1).astype(np.float32)
            pose = torch.from_numpy(pose).cuda()
            us = timed(lambda: P.crop_and_com_from_pose(dms, pose, cfg, 128, 128, dataset=name))
            crops, _, ncfg, com = P.crop_and_com_from_pose(dms, pose, cfg, 128, 128, dataset=name)
            byts = B * (160 * 160 + 128 * 128) * 4.0            # ~160x160 box read + 128x128 crop written per frame
            print('| crop + com | %d | %s %dx%d | %.1f | %.3g | %.0f |' % (B, name, H, W, us, B / us * 1e6, byts / us / 1e3))
            draws = torch.from_numpy(P.draw_aug_params(B, rng)).cuda()
            us = timed(lambda: P.data_aug(crops, pose, ncfg, com, draws))
            print('| data_aug | %d | 128x128 crops | %.1f | %.3g | %.0f |' % (B, us, B / us * 1e6, B * 2 * 128 * 128 * 4.0 / us / 1e3))


if __name__ == '__main__':
    main()
