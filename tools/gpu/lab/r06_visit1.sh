#!/usr/bin/env bash
# round 6, visit 1: conv_p3.h (x3 products on a P3-stored input, all LDS-DMA, three-stage ring) -- first hardware run:
# bit-identity with conv_x3_kernel on the GPU, then time per launch against it, shape by shape at 200 crops
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_forward_parity.py -q -m gpu -k "conv_x3" -p no:cacheprovider -x > gpurun_out/r06v1_p3_parity.log 2>&1; echo "rc=$?" >> gpurun_out/r06v1_p3_parity.log
grep -v "start\]\|passed\]" gpurun_out/r06v1_p3_parity.log | tail -15
timeout 900 python tools/x3_bench.py 200 > gpurun_out/r06v1_x3_bench.md 2> gpurun_out/r06v1_x3_bench.err; tail -3 gpurun_out/r06v1_x3_bench.err
cut -c1-200 gpurun_out/r06v1_x3_bench.md
