#!/usr/bin/env bash
# round 3, visit 37: SQ counters of the dominant conv kernel at the window pass's launch shape (3x3 256->256, 200 crops, 64x128 tile)
export PROBE_B=200
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash tools/gpu/conv_sq_counters.sh > gpurun_out/v37_conv_counters_b200.log 2>&1
cp gpurun_out/conv_counters.md gpurun_out/v37_conv_counters_b200.md
cat gpurun_out/v37_conv_counters_b200.md | head -60
