#!/usr/bin/env bash
# round 6, visit 2: where conv_p3_kernel's time goes -- DR_P3_VARIANT 0 (product) | 1 copies issued between the MFMA groups | 2 no copies in
# the loop | 4 no waits / barriers | 6 neither
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in 0 1 2 4 6 0 1; do DR_P3_VARIANT=$v timeout 300 python tools/p3_bench.py 200 2>/dev/null | cut -c1-110; done | tee gpurun_out/r06v2_p3_variants.md
