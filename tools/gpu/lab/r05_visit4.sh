#!/usr/bin/env bash
# round 5, visit 4: the driver's pytest command on a fresh box (new tests included), then the tests that print measurements, with -s
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -f gpurun_out/pytest_live.log
( time timeout 1100 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/v4_suite.log 2>&1; echo "rc=$?" >> gpurun_out/v4_suite.log
cp gpurun_out/pytest_live.log gpurun_out/v4_pytest_live.log
timeout 900 python -m pytest tests/test_trained_parity.py tests/test_gpu_configs.py tests/test_bench_shapes.py tests/test_train_parity.py -q -m gpu -s -p no:cacheprovider > gpurun_out/v4_measure.log 2>&1; echo "rc=$?" >> gpurun_out/v4_measure.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/v4_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/v4_smoke.log
grep -v "start\]\|passed\]" gpurun_out/v4_suite.log | tail -15; tail -2 gpurun_out/v4_smoke.log
grep "grad vs\|held-out\|trained weights\|loss terms\|ReplicaPool\|rc=\|passed\|failed" gpurun_out/v4_measure.log | grep -v "passed\]" | cut -c1-330
