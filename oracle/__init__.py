"""CPU oracle for the denseReg hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (PyTorch-CPU fp32/fp64 + numpy) of the
reference algorithm for the one hot path this repo accelerates: the
stacked-hourglass network of ``network/um_v1.py`` (forward, train-mode
BatchReNorm, loss, gradients), and the offset vote of
``model/hourglass_um_crop_tiny.py:743-785``.

PARITY UNPINNED: the reference is Python-2.7 + TensorFlow-1.3 and can be neither
imported nor compiled in this image, it ships no tests and
no golden vectors for this path.  The restatement follows the cited reference
lines, is pinned only by hand-derived known-answer tests (tests/test_oracle_*.py)
and by the format of ``exp/result/*.txt``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker.  Nothing under
``densereg_amd/`` imports it; the product path fails loudly when the HIP
library is missing.
"""
from .graph import NetConfig, conv_specs, param_specs, trainable_names  # noqa: F401
