import os
import sys
import time
import warnings

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

warnings.filterwarnings('ignore', message='Converting a tensor with requires_grad')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # pytest.ini's timeout / timeout_method keys and the per-test limits below belong to pytest-timeout (part of this image, listed in
    # tests/requirements.txt); without the plugin the marker is registered here so --strict-markers still passes and no limit applies
    if not config.pluginmanager.hasplugin('timeout'):
        config.addinivalue_line('markers', 'timeout(seconds): per-test limit (inactive: pytest-timeout is not installed)')
        warnings.warn('pytest-timeout is not installed: the per-test time limits of pytest.ini / conftest.py are inactive')
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


# ---- order: the oracle-parity tests of the hot path first, executor-mode / reproducibility / stress loops last --------------------
# (a run that is cut short -- the driver gives `pytest -m gpu` a wall-clock limit -- has then seen the parity evidence)
_FILE_ORDER = ['test_abi.py', 'test_oracle_known_answers.py', 'test_gpu_configs.py', 'test_gpu_fullsize.py', 'test_trained_parity.py',
               'test_train_parity.py', 'test_bench_shapes.py', 'test_forward_parity.py', 'test_bn_layer.py', 'test_pool.py',
               'test_fused_tail.py', 'test_frontend.py', 'test_dataio.py', 'test_checkpoint.py', 'test_host_mirror.py',
               'test_groups.py', 'test_data_parallel.py', 'test_configs_emu.py', 'test_kernel_resources.py', 'test_pipeline.py']


def pytest_collection_modifyitems(config, items):
    rank = {f: i for i, f in enumerate(_FILE_ORDER)}
    pos = {id(it): i for i, it in enumerate(items)}

    def key(it):
        return (rank.get(os.path.basename(str(it.fspath)), len(_FILE_ORDER) - 1), pos[id(it)])
    items.sort(key=key)
    have_timeout = config.pluginmanager.hasplugin('timeout')
    for it in items:                      # a GPU test that stalls is cut (and named, with every thread's stack) long before the suite's wall-clock limit
        if have_timeout and it.get_closest_marker('gpu') is not None and it.get_closest_marker('timeout') is None:
            it.add_marker(pytest.mark.timeout(240))


# ---- one flushed line per test: the tail of a log that was cut names the last finished test and the running one -------------------
def _live(config, line):
    tr = config.pluginmanager.get_plugin('terminalreporter')
    if tr is not None and os.environ.get('DR_TEST_LIVE', '1') != '0':
        tr.ensure_newline()
        tr.write_line(line)
        try:
            tr._tw.flush()
        except Exception:
            pass
    try:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'pytest_live.log'), 'a') as f:
            f.write(line + '\n')
    except OSError:
        pass


@pytest.hookimpl(tryfirst=True)
def pytest_runtest_logstart(nodeid, location):
    cfg = _live.config
    if cfg is not None:
        _live(cfg, '[%s start] %s' % (time.strftime('%H:%M:%S'), nodeid))


def pytest_runtest_logreport(report):
    cfg = _live.config
    if cfg is None:
        return
    if report.when == 'call' or (report.when == 'setup' and report.outcome != 'passed'):
        _live(cfg, '[%s %s] %s %.2fs' % (time.strftime('%H:%M:%S'), report.outcome, report.nodeid, report.duration))


_live.config = None


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
    _live.config = session.config
    # only the lines THIS session appends are summarised
    try:
        session.config._dr_branch_mark = os.path.getsize(os.path.join(ROOT, 'gpurun_out', 'test_branches.jsonl'))
    except OSError:
        session.config._dr_branch_mark = 0
