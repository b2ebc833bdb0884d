#!/usr/bin/env bash
# round 2, visit 1: all gpu tests (new config 3/4/5 + BatchReNorm layer pins), smoke, default bench with per-layer detail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=15 > gpurun_out/r02_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r02_smoke.log
timeout 900 python bench.py --detail gpurun_out/r02_detail_train.md > gpurun_out/r02_bench_train.json 2> gpurun_out/r02_bench_train.err; echo "bench rc=$?" >> gpurun_out/r02_bench_train.err
tail -25 gpurun_out/r02_pytest_gpu.log; tail -2 gpurun_out/r02_smoke.log; cut -c1-400 gpurun_out/r02_bench_train.json; tail -3 gpurun_out/r02_bench_train.err
nproc; free -g | head -2
