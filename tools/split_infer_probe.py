#!/usr/bin/env python
"""forward(eval)+vote of ONE batch of 40 crops as k concurrent sub-batches (k engines of max_batch 40/k on k streams, joined per
step) vs one engine at B=40: eval-mode BatchReNorm has no batch coupling, so the split is exact.  python tools/split_infer_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densereg_amd.data.synthetic import DATASETS, make_crops  # noqa: E402
from densereg_amd.engine import Engine  # noqa: E402
from tools.dual_stream_probe import params_for  # noqa: E402


def main():
    steps = 60
    dev = torch.device('cuda', 0)
    S, F, J = 2, 128, DATASETS['icvl']['jnt_num']
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for B in (40, 8, 1):
        for k in (1, 2, 4):
            if B % k:
                continue
            sub = B // k
            engs = [Engine(S, F, J, 128, 3, sub, 0, training=False) for _ in range(k)]
            for e in engs:
                e.load_params(params_for(e))
            dm, poses, cfgs, coms, _ = make_crops(B, 'icvl', seed=20240)
            ndm = engs[0].norm_dm(t(dm[:sub]), t(coms[:sub]))
            d_dm = [engs[i].norm_dm(t(dm[i * sub:(i + 1) * sub]), t(coms[i * sub:(i + 1) * sub])) for i in range(k)]
            d_cfg = [t(cfgs[i * sub:(i + 1) * sub]) for i in range(k)]
            d_com = [t(coms[i * sub:(i + 1) * sub]) for i in range(k)]
            xyz = [e.new(sub, 3 * J) for e in engs]
            streams = [torch.cuda.Stream(dev) for _ in range(k)]
            main_s = torch.cuda.current_stream(dev)
            evs = [torch.cuda.Event() for _ in range(k)]

            def step():
                e0 = torch.cuda.Event(); e0.record(main_s)
                for i in range(k):
                    streams[i].wait_event(e0)
                    with torch.cuda.stream(streams[i]):
                        engs[i].infer(d_dm[i], d_cfg[i], d_com[i], out=xyz[i])
                        evs[i].record(streams[i])
                for i in range(k):
                    main_s.wait_event(evs[i])
            for _ in range(10):
                step()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
            print('B=%d as %d x %d: %.1f crops/s, %.3f ms per batch' % (B, k, sub, B * steps / el, el / steps * 1e3))
            sys.stdout.flush()
            for e in engs:
                e.close()


if __name__ == '__main__':
    main()
