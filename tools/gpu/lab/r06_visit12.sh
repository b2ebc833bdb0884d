#!/usr/bin/env bash
# round 6, visit 12: executor lanes (hourglass branches on their own streams) at the 200-crop window, A/B
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
Q="--steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > gpurun_out/r06v12_$name.json 2> gpurun_out/r06v12_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/r06v12_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)"; }
run base_1 A=1
run lanes_1 DR_MULTI_STREAM=1
run base_2 A=1
run lanes_2 DR_MULTI_STREAM=1
run bngrid768 DR_BN_GRID=768
run bngrid384 DR_BN_GRID=384
run wgflush3 DR_WGRAD_STREAM=3
run base_3 A=1
