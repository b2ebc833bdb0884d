#!/usr/bin/env bash
# GPU visit: host-mirror / checkpoint / front-end tests
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_host_mirror.py tests/test_checkpoint.py tests/test_frontend.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_host.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_host.log
tail -25 gpurun_out/pytest_host.log
