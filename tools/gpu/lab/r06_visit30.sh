#!/usr/bin/env bash
# round 6, visit 30: what the two-stage loop of conv_x3_kernel spends on what (DR_X3_ABL: 3 no stores, 16 no pixel path, 32 no weight copies, 64 no waits / barrier)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for a in 0 3 16 32 48 51 112 115 0; do DR_X3_ABL=$a timeout 120 python tools/x3_intercept_bench.py 200 2>/dev/null | sed -n '1p;3,5p'; done | tee gpurun_out/r06v30_abl.md
