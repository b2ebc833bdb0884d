"""Identity of the kernel sources a measurement was taken with.

``profiles/pmc_traffic.json`` (HBM bytes per launch from rocprofv3 PMC passes) is produced outside the benchmark process;
``bench.py`` may only quote it while the kernels it describes are the kernels of the build it is running.  The stamp is a hash
of the kernel sources and of the launch-side sources (tile selection, executors) with comments and whitespace removed, so an edit to a comment does not invalidate a measurement and an
edit to the code does.
"""
from __future__ import annotations

import hashlib
import os
import re

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
# the kernels, and the host code that decides WHICH kernel runs with what grid (tile selection, slab plans, split-K fallbacks,
# the executors): an edit there changes the bytes per launch just as an edit to a kernel does
KERNEL_SOURCES = ('conv_igemm.h', 'conv_x3.h', 'conv_x3h.h', 'conv_p3.h', 'conv_wgrad_x3.h', 'conv_epilogue.inc', 'conv_splitk.h', 'conv_wgrad.h', 'conv_wgrad16.h', 'conv_wgrad_bf16.h', 'conv_wgrad_tr.h', 'hg_fused.h',
                  'train_kernels.h', 'kernels_misc.h', 'dr_platform.h', 'vote.h',
                  'densereg.cpp', 'train_exec.inc', 'pipeline.inc', 'net.h')


def _strip(src: str) -> str:
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    src = re.sub(r'//[^\n]*', '', src)
    return re.sub(r'\s+', '', src)


def kernel_source_hash(files=KERNEL_SOURCES) -> str:
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(_CSRC, f)) as fh:
            h.update(f.encode() + b'\0' + _strip(fh.read()).encode() + b'\0')
    return h.hexdigest()[:16]


if __name__ == '__main__':
    print(kernel_source_hash())
