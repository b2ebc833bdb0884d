#!/usr/bin/env bash
# round 3, visit 40: the 128x128 conv tile with the LDS-DMA refill (no staging registers -> four workgroups per CU at 128 VGPRs, or
# three at 162) against the register-staged one and the 64x128 tile, 200 and 40 crops per launch
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
S="32:256:256:3:1 32:256:256:3:0 32:512:512:1:1 32:512:512:1:0 32:515:512:1:0 32:128:128:3:1 32:128:128:3:0 32:256:512:1:0 32:512:256:1:1 32:512:256:1:0 32:128:256:1:0 32:256:128:1:0"
echo "== register-staged 128x128 (tile 0), 200 crops"; PROBE_B=200 timeout 600 python tools/conv_probe.py $S 2>&1 | tail -12
echo "== LDS-DMA 128x128 (tile 0), 200 crops"; DR_CONV_GLDS128=1 PROBE_B=200 timeout 600 python tools/conv_probe.py $S 2>&1 | tail -12
echo "== LDS-DMA 128x128, 40 crops"; DR_CONV_GLDS128=1 PROBE_B=40 timeout 600 python tools/conv_probe.py 32:256:256:3:1 32:256:256:3:0 32:512:512:1:1 32:512:512:1:0 2>&1 | tail -4
echo "== register-staged, 40 crops"; PROBE_B=40 timeout 600 python tools/conv_probe.py 32:256:256:3:0 32:512:512:1:0 2>&1 | tail -2
