#!/usr/bin/env bash
# round 3, visit 26: micro-batch groups with the executor lanes (DR_MULTI_STREAM=1: the hourglass branches on their own streams) --
# at 200 crops per pass the branches are no longer launch-bound; parity first, then the training step
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
DR_MULTI_STREAM=1 timeout 600 python -m pytest tests/test_groups.py -m gpu -x -q 2>&1 | tail -5
T="--no-cpu-baseline --no-profile --no-forward-vote --steps 100 --warmup 10"
run() { name=$1; shift; env "$@" > $G/v26_$name.json 2> $G/v26_$name.err; python -c "
import json;d=json.load(open('$G/v26_$name.json'));print('$name',round(d['value'],1),round(d['ms_per_step'],3))" 2>/dev/null || { echo "$name FAILED"; tail -5 $G/v26_$name.err; }; }
run g5 timeout 300 python bench.py $T
run g5_lanes DR_MULTI_STREAM=1 timeout 300 python bench.py $T
run g5_bn1024 DR_BN_GRID=1024 timeout 300 python bench.py $T
run g5_bn2048 DR_BN_GRID=2048 timeout 300 python bench.py $T
run g5_red512 DR_BN_RED_GRID=512 timeout 300 python bench.py $T
run g5_nofuse DR_FUSE_BN_BWD=0 timeout 300 python bench.py $T
run g5_nogroupwg DR_GROUP_WGRAD=0 timeout 300 python bench.py $T
