#!/usr/bin/env bash
# round 3, visit 29: where the 64x128 tile's time goes at 200 crops per launch (ablations of dr_dbg_conv_bench), 3x3 256->256 and
# 1x1 512->512 (128x128 tile: ablations 1-3)
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
PROBE_B=200 timeout 600 python tools/conv_probe.py 32:256:256:3:1:0 32:256:256:3:1:7 32:256:256:3:1:8 32:256:256:3:1:9 32:256:256:3:1:10 \
  32:256:256:3:1:11 32:256:256:3:1:12 32:256:256:3:1:13 32:256:256:3:1:6 \
  32:512:512:1:0:0 32:512:512:1:0:1 32:512:512:1:0:3 32:512:512:1:1:0 32:512:512:1:1:7 32:512:512:1:1:12 32:512:512:1:1:13 \
  32:128:128:3:1:0 32:128:128:3:1:7 32:128:128:3:1:12 32:128:128:3:1:13 > $G/v29_ablations_b200.md 2>&1
cat $G/v29_ablations_b200.md
