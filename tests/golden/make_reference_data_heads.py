#!/usr/bin/env python
"""Copy the first records of the DATA files the reference ships into tests/golden/ (run in the dev container, where
/root/reference is mounted; the GPU box and the tests only see the committed copies).

* ``exp/result/icvl.txt`` / ``nyu.txt``  -> ``icvl_result_head.txt`` / ``nyu_result_head.txt``: pin the per-frame output FORMAT
  (name, tab, %.4f fields, backslash separators, joint count) -- they are predictions of the trained model on the real
  test sets, so they say nothing numeric about the kernels.
* ``data/nyu_bbx.pkl`` (Python-2 pickle: list of 8252 float32 arrays of shape (5, 1)) -> ``nyu_bbx_head.npy`` (16 rows):
  pins the (top, left, bottom, right, depth threshold) convention ``crop_from_bbx`` consumes (data/preprocess.py:81-129).

    python tests/golden/make_reference_data_heads.py
"""
import os
import pickle

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def main():
    for name in ('icvl', 'nyu'):
        with open(os.path.join(REF, 'exp/result/%s.txt' % name)) as f:
            head = [next(f) for _ in range(3)]
        open(os.path.join(HERE, '%s_result_head.txt' % name), 'w').writelines(head)
    with open(os.path.join(REF, 'data/nyu_bbx.pkl'), 'rb') as f:
        boxes = np.asarray(pickle.load(f, encoding='latin1'), np.float32).reshape(-1, 5)
    np.save(os.path.join(HERE, 'nyu_bbx_head.npy'), boxes[:16])


if __name__ == '__main__':
    main()
