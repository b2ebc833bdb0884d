#!/usr/bin/env bash
# round 3, visit 22: config 5 (S=4 F=256 256x256, bf16): BatchReNorm grid caps / rows in flight at 4x the tensor sizes
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
C5="--num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu --no-cpu-baseline --no-profile --no-forward-vote --steps 10 --warmup 3 --precision bf16"
run() { name=$1; shift; env "$@" timeout 400 python bench.py $C5 > $G/v22_$name.json 2> $G/v22_$name.err; python -c "
import json;d=json.load(open('$G/v22_$name.json'));print('$name',round(d['value'],1),round(d['ms_per_step'],3))" 2>/dev/null || { echo "$name FAILED"; tail -5 $G/v22_$name.err; }; }
run base DR_PIPELINE=1
run bn1024 DR_BN_GRID=1024
run bn2048 DR_BN_GRID=2048
run bn4096 DR_BN_GRID=4096
run red512 DR_BN_RED_GRID=512
run red1024 DR_BN_RED_GRID=1024
run bn2048red1024 DR_BN_GRID=2048 DR_BN_RED_GRID=1024
run nows DR_WGRAD_STREAM=0
