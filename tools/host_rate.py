#!/usr/bin/env python
"""Host cost of enqueueing one step with an empty GPU queue (sync before every step): what the launch loop itself takes.
python tools/host_rate.py [train|infer]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from densereg_amd.engine import Engine  # noqa: E402
from densereg_amd.parallel import DataParallelTrainer  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'train'
    dataset = 'nyu' if mode == 'train' else 'icvl'
    J = bench.DATASETS[dataset]['jnt_num']
    B = 40
    dev = torch.device('cuda', 0)
    eng = Engine(2, 128, J, 128, 3, B, 0, training=(mode == 'train'))
    rng = np.random.default_rng(7)
    params = {}
    for name, shape, _ in eng.param_infos():
        leaf = name.rsplit('/', 1)[1]
        if leaf == 'weights':
            params[name] = (rng.standard_normal(shape) * np.sqrt(2.0 / (shape[0] * shape[1] * shape[2]))).astype(np.float32)
        elif leaf in ('gamma', 'moving_variance', 'r_max'):
            params[name] = np.ones(shape, np.float32)
        else:
            params[name] = np.zeros(shape, np.float32)
    eng.load_params(params)
    dm, poses, cfgs, coms, _ = bench.make_crops(B, dataset, seed=20240, rank=0, hw=128)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_dm_mm, d_pose, d_cfg, d_com = t(dm), t(poses), t(cfgs), t(coms)
    d_dm = eng.norm_dm(d_dm_mm, d_com)
    xyz = eng.new(B, 3 * J)
    trainer = DataParallelTrainer(eng, dataset=dataset, sub_batch=5, dist=None) if mode == 'train' else None

    def step(i):
        if mode == 'infer':
            eng.infer(d_dm, d_cfg, d_com, out=xyz)
        else:
            trainer.micro_step(d_dm, d_pose, d_cfg, d_com, seed=i)

    for i in range(6):
        step(i)
    host, total = [], []
    for i in range(20):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        step(10 + i)
        t1 = time.perf_counter()
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3)
        total.append((t2 - t0) * 1e3)
    print('%s: host enqueue %.2f ms (min %.2f), step with empty queue %.2f ms' % (mode, float(np.median(host)), min(host), float(np.median(total))))


if __name__ == '__main__':
    main()
