"""The N > 1 host logic of ``bench.py`` without a GPU (``DR_BENCH_DRYRUN=1``): ``spawn_ranks`` -> ``torch.distributed.run`` with two
ranks on 127.0.0.1, the launcher's RANK / LOCAL_RANK / WORLD_SIZE, rank -> device binding, the process group (gloo standing in for
RCCL), ``DataParallelTrainer.window_step`` with its all-reduce of the flat gradient (CHECKED element by element by the engine
stand-in), barrier + MAX-over-ranks timing, and the contract of the output: exactly one JSON line, from rank 0, whole-job value.
No kernel runs and nothing is measured -- what this guards is that the first real multi-GPU launch (the driver's, on an 8-GPU node)
cannot die on host logic (reference: model/train_multi_gpu.py:56-92, the tower loop this replaces)."""
import json
import os
import subprocess
import sys

from tests.common import ROOT


def _run(argv, env_extra, timeout=240):
    env = dict(os.environ, DR_BENCH_DRYRUN='1', **env_extra)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        if k not in env_extra:
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + argv, env=env, capture_output=True, text=True, timeout=timeout)


def test_two_ranks_window_steps_all_reduce_and_one_json_line():
    r = _run(['--gpus', '2', '--steps', '10', '--warmup', '5'], {})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                      # rank 0 alone, and nothing else on stdout
    d = json.loads(lines[0])
    assert d['dry_run'] is True and d['metric'].startswith('DRY RUN')
    assert d['n_gpus'] == 2 and d['steps'] == 10 and d['warmup'] == 5 and d['scaling'] == 'weak' and d['higher_is_better'] is True
    cfg = d['config']
    assert cfg['world_size'] == 2 and cfg['parallelism'] == 'dp2' and cfg['global_batch'] == 80 and cfg['micro_steps_per_pass'] == 5
    # 15 micro-steps = 3 windows: every optimizer step saw the SUM over both ranks on every element of the flat gradient
    assert cfg['checked_all_reduces'] == 3
    # whole-job value = crops of ALL ranks / the slowest rank's time
    assert abs(d['value'] - 40 * 2 * 10 / (d['ms_per_step'] * 10 / 1e3)) < 1e-6 * d['value']
    # each rank bound its own device
    assert 'rank 0/2: LOCAL_RANK=0 -> pretend device 0 of 2' in r.stderr and 'rank 1/2: LOCAL_RANK=1 -> pretend device 1 of 2' in r.stderr


def test_narrowed_device_visibility_binds_device_zero_on_every_rank():
    r = _run(['--gpus', '2', '--steps', '5', '--warmup', '0'], {'DR_BENCH_DRYRUN_DEVICES': '1'})
    assert r.returncode == 0, r.stderr[-2000:]
    assert 'rank 1/2: LOCAL_RANK=1 -> pretend device 0 of 1' in r.stderr
    assert json.loads(r.stdout.strip())['config']['checked_all_reduces'] == 1


def test_rank_count_that_differs_from_gpus_is_refused():
    r = _run(['--gpus', '2', '--steps', '1'], {'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode == 2 and 'they must agree' in r.stderr
    assert r.stdout.strip() == ''
