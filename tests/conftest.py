import os
import sys
import warnings

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

warnings.filterwarnings('ignore', message='Converting a tensor with requires_grad')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the CPU oracle: at most 32 threads (the 256-thread GPU host thrashes with torch's default of one thread per core)
    try:
        import torch
        torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    except Exception:
        pass


@pytest.fixture(scope='session')
def emu():
    from tests.common import EmuBackend
    return EmuBackend()


@pytest.fixture(scope='session')
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail('-m gpu tests need a GPU: the HIP path has no fallback')
    from tests.common import GpuBackend
    return GpuBackend()
