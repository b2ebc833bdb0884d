#!/usr/bin/env bash
# GPU visit: split-K conv kernel -- parity tests and the small-layer micro-benchmark
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_forward_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "splitk or every_tile or bf16_matrix" 2>&1 | tail -4
timeout 300 python tools/conv_small_bench.py > gpurun_out/conv_small_bench.md 2> gpurun_out/conv_small_bench.err; cat gpurun_out/conv_small_bench.md; tail -2 gpurun_out/conv_small_bench.err
