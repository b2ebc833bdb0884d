#!/usr/bin/env bash
# round 4 visit 19: workgroup caps and side-stream settings re-swept at the window shape (200 crops per launch), fp32 headline
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; G=gpurun_out
b() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > $G/r04_v19_$name.json 2> $G/r04_v19_$name.err; python -c "import json; d=json.load(open('$G/r04_v19_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))"; }
Q="--no-cpu-baseline --no-forward-vote --steps 40 --warmup 10 --no-profile"
b base A=1
b bn_grid_320 DR_BN_GRID=320
b bn_grid_768 DR_BN_GRID=768
b bn_grid_1280 DR_BN_GRID=1280
b bn_grid_2560 DR_BN_GRID=2560
b red_128 DR_BN_RED_GRID=128
b red_512 DR_BN_RED_GRID=512
b red_1024 DR_BN_RED_GRID=1024
b base2 A=1
b elt_1024 DR_ELT_GRID=1024
b elt_4096 DR_ELT_GRID=4096
b wgs_2 DR_WGRAD_STREAM=2
b wgs_4 DR_WGRAD_STREAM=4
b wgprio0 DR_WG_PRIO=0
b nfast0 DR_CONV_NFAST=0
b base3 A=1
