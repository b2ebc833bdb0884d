#!/usr/bin/env bash
# GPU visit: train parity tests + train bench with per-layer detail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_train_parity.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_train.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_train.log
timeout 900 python bench.py --steps 20 --warmup 5 --detail gpurun_out/detail_train.md --no-cpu-baseline > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err; echo "bench rc=$?" >> gpurun_out/bench_train.err
tail -3 gpurun_out/pytest_train.log; cut -c1-330 gpurun_out/bench_train.json; echo
