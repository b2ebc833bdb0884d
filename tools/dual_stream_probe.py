#!/usr/bin/env python
"""Upper bound of what pipelining micro-steps could buy: TWO independent engines (two handles, two torch streams) run the
bench's training micro-step concurrently on one GPU; aggregate crops/s vs one engine alone.  Kernels of the two streams fill each
other's launch boundaries and small-grid chains.  (Also the forward(eval)+vote engine, two replicas.)

    python tools/dual_stream_probe.py [steps]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densereg_amd.data.synthetic import DATASETS, make_crops  # noqa: E402
from densereg_amd.engine import Engine  # noqa: E402
from densereg_amd.parallel import DataParallelTrainer  # noqa: E402


def params_for(e):
    rng = np.random.default_rng(7)
    out = {}
    for name, shape, _ in e.param_infos():
        leaf = name.rsplit('/', 1)[1]
        if leaf == 'weights':
            out[name] = (rng.standard_normal(shape) * np.sqrt(2.0 / (shape[0] * shape[1] * shape[2]))).astype(np.float32)
        elif leaf in ('gamma', 'moving_variance', 'r_max'):
            out[name] = np.ones(shape, np.float32)
        else:
            out[name] = np.zeros(shape, np.float32)
    return out


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device('cuda', 0)
    B, S, F = 40, 2, 128
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for mode in ('train', 'infer'):
        ds = 'nyu' if mode == 'train' else 'icvl'
        J = DATASETS[ds]['jnt_num']
        for n_eng in (1, 2, 3):
            engs, trs, data, streams = [], [], [], []
            for r in range(n_eng):
                e = Engine(S, F, J, 128, 3, B, 0, training=(mode == 'train'))
                e.load_params(params_for(e))
                dm, poses, cfgs, coms, _ = make_crops(B, ds, seed=20240 + r)
                d = (e.norm_dm(t(dm), t(coms)), t(poses), t(cfgs), t(coms))
                engs.append(e); data.append(d); streams.append(torch.cuda.Stream(dev))
                trs.append(DataParallelTrainer(e, dataset=ds, sub_batch=5) if mode == 'train' else None)
            xyz = [e.new(B, 3 * J) for e in engs]
            torch.cuda.synchronize(dev)

            def step(i):
                for r in range(n_eng):
                    with torch.cuda.stream(streams[r]):
                        if mode == 'train':
                            trs[r].micro_step(*data[r], seed=i)
                        else:
                            engs[r].infer(data[r][0], data[r][2], data[r][3], out=xyz[r])
            for i in range(8):
                step(i)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(steps):
                step(8 + i)
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
            print('%s: %d concurrent engine(s): %.1f crops/s aggregate (%.3f ms per step of each)' % (mode, n_eng, B * n_eng * steps / el, el / steps * 1e3))
            sys.stdout.flush()
            for e in engs:
                e.close()


if __name__ == '__main__':
    main()
