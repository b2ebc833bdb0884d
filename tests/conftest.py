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


def pytest_terminal_summary(terminalreporter):
    """Which gradient bar the training-step tests applied on THIS box (tests/test_train_parity.py::_record_branch: the fp64
    autograd of the oracle needs host RAM; ``gpurun_out/`` does not travel back from the driver's box, the log does)."""
    import json
    path = os.path.join(ROOT, 'gpurun_out', 'test_branches.jsonl')
    mark = getattr(terminalreporter.config, '_dr_branch_mark', 0)
    try:
        with open(path) as f:
            f.seek(mark)
            lines = [ln.strip() for ln in f if ln.strip()]
    except OSError:
        return
    if not lines:
        return
    terminalreporter.write_sep('-', 'gradient bars applied (test_branches.jsonl)')
    for ln in lines:
        try:
            d = json.loads(ln)
            terminalreporter.write_line('%s: %s  (B=%s S=%s F=%s J=%s)' % (d.get('test', '?').split(' ')[0], d.get('gradient_bar'), d.get('B'),
                                                                        d.get('S'), d.get('F'), d.get('J')))
        except ValueError:
            terminalreporter.write_line(ln)


def pytest_sessionstart(session):
    # only the lines THIS session appends are summarised
    try:
        session.config._dr_branch_mark = os.path.getsize(os.path.join(ROOT, 'gpurun_out', 'test_branches.jsonl'))
    except OSError:
        session.config._dr_branch_mark = 0
