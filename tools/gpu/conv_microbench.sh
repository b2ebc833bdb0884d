#!/usr/bin/env bash
# GPU visit: conv microbench only (+ gpu tests of the conv tile sweep)
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python tools/conv_bench.py > gpurun_out/conv_bench.md 2> gpurun_out/conv_bench.err
timeout 600 python -m pytest tests/test_forward_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "tile" > gpurun_out/pytest_tile.log 2>&1
tail -2 gpurun_out/pytest_tile.log; cat gpurun_out/conv_bench.md
