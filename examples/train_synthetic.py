#!/usr/bin/env python
"""Train the S=2 F=128 network on learnable synthetic hands with the engine and report the held-out joint error as it falls.

    python examples/train_synthetic.py [--steps 300] [--crops 2000] [--dataset icvl]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densereg_amd.data.synthetic import make_hand_crops  # noqa: E402
from densereg_amd.synthetic_training import evaluate, joint_error_mm, reference_init, train  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--crops', type=int, default=2000)
    ap.add_argument('--dataset', default='icvl')
    ap.add_argument('--num_stack', type=int, default=2)
    ap.add_argument('--num_fea', type=int, default=128)
    ap.add_argument('--save', default='')
    a = ap.parse_args()
    held = make_hand_crops(200, a.dataset, seed=991)[:4]
    t0 = time.time()
    params, hist = train(a.num_stack, a.num_fea, a.dataset, a.steps, a.crops, log=print)
    print('trained %d steps in %.1f s (incl. crop synthesis)' % (a.steps, time.time() - t0))
    xyz = evaluate(params, a.num_stack, a.num_fea, a.dataset, held)
    e = joint_error_mm(xyz, held[1])
    print('held-out synthetic hands: mean joint error %.2f mm, median %.2f, max %.1f' % (e.mean(), np.median(e), e.max()))
    if a.save:
        np.savez(a.save, **params)


if __name__ == '__main__':
    main()
