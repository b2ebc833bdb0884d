#!/usr/bin/env bash
# round 3, visit 35: the window pass on the bf16 matrix cores against the micro-step loop
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_groups.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -12
