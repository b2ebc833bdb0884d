#!/usr/bin/env bash
# GPU visit: front-end tests + throughput table
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_frontend.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_frontend.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_frontend.log
timeout 300 python tools/frontend_bench.py > gpurun_out/frontend_bench.md 2> gpurun_out/frontend_bench.err
tail -5 gpurun_out/pytest_frontend.log; cat gpurun_out/frontend_bench.md; tail -3 gpurun_out/frontend_bench.err
