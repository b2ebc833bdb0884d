#!/usr/bin/env bash
# round 3, visit 21: executor switches re-measured under the two-slot pipeline
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
Q="--no-cpu-baseline --no-profile --no-forward-vote --steps 50 --warmup 10"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > $G/v21_$name.json 2> $G/v21_$name.err; python -c "
import json;d=json.load(open('$G/v21_$name.json'));print('$name',round(d['value'],1),round(d['ms_per_step'],3))" 2>/dev/null || { echo "$name FAILED"; tail -5 $G/v21_$name.err; }; }
run base DR_PIPELINE=2
run ws2 DR_WGRAD_STREAM=2
run ws3 DR_WGRAD_STREAM=3
run ws5 DR_WGRAD_STREAM=5
run ws9 DR_WGRAD_STREAM=9
run nofuse_act DR_FUSE_ACT=0
run nofuse_last DR_FUSE_LAST=0
run red512 DR_BN_RED_GRID=512
run red128 DR_BN_RED_GRID=128
run nfast0 DR_CONV_NFAST=0
run glds0 DR_CONV_GLDS=0
run base2 DR_PIPELINE=2
