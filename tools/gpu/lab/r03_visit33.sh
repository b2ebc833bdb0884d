#!/usr/bin/env bash
# round 3, visit 33: a 256x128 tile (two workgroups per CU, half the operand traffic per flop of the 64x128 tile) at 200 crops
# per launch, against the heuristic's tile; parity of the tile on one shape through the conv test hook
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
PROBE_B=200 timeout 600 python tools/conv_probe.py 32:256:256:3:-1 32:256:256:3:28 32:512:512:1:-1 32:512:512:1:28 32:515:512:1:28 \
  32:128:128:3:-1 32:128:128:3:28 32:256:512:1:-1 32:256:512:1:28 32:512:256:1:-1 32:512:256:1:28 32:128:256:1:-1 32:128:256:1:28 \
  32:256:128:1:-1 32:256:128:1:28 > $G/v33_tile256.md 2>&1
cat $G/v33_tile256.md
DR_CONV_GLDS=0 PROBE_B=200 timeout 600 python tools/conv_probe.py 32:256:256:3:28 32:512:512:1:28 2>&1 | tail -2
PROBE_B=40 timeout 600 python tools/conv_probe.py 32:256:256:3:-1 32:256:256:3:28 32:512:512:1:-1 32:512:512:1:28 2>&1 | tail -4
