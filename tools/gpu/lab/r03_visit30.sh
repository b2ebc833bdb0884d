#!/usr/bin/env bash
# round 3, visit 30: weight-gradient slab count around the plan's choice, big layers, at 40 and 200 crops per launch
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export PROBE_SHAPES="32:256:256:3,32:128:128:3,32:512:512:1,32:515:512:1,32:256:512:1,32:512:256:1,32:128:256:1,32:256:128:1,32:64:64:3"
export PROBE_NS="0,12,16,20,24,28,32,40,48,56,64,72,80,96,112,128,160,192,256" PROBE_T=128
PROBE_B=200 timeout 900 python tools/wgrad_bench.py > $G/v30_wgrad_ns_b200.md 2>&1
PROBE_B=40 timeout 900 python tools/wgrad_bench.py > $G/v30_wgrad_ns_b40.md 2>&1
PROBE_T=64 PROBE_SHAPES="32:64:64:3,32:128:64:1,32:64:128:1,16:64:64:3" PROBE_B=200 timeout 900 python tools/wgrad_bench.py > $G/v30_wgrad_ns64_b200.md 2>&1
wc -l $G/v30_*.md
