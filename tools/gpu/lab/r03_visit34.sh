#!/usr/bin/env bash
# round 3, visit 34: executor lanes in the FORWARD passes only (the hourglass's upper residual beside the pooled pyramid), window pass
# and merged inference
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
DR_FWD_LANES=1 timeout 600 python -m pytest tests/test_groups.py tests/test_train_parity.py -m gpu -x -q -k "groups or single_stack or config3" 2>&1 | tail -3
T="--no-cpu-baseline --no-profile --no-forward-vote --steps 100 --warmup 10"
run() { name=$1; shift; env "$@" > $G/v34_$name.json 2> $G/v34_$name.err; python -c "
import json;d=json.load(open('$G/v34_$name.json'));print('$name',round(d['value'],1),round(d['ms_per_step'],3),(d['config'].get('single_replica') or {}).get('value'))" 2>/dev/null || { echo "$name FAILED"; tail -5 $G/v34_$name.err; }; }
run base timeout 300 python bench.py $T
run fwd_lanes DR_FWD_LANES=1 timeout 300 python bench.py $T
run base2 timeout 300 python bench.py $T
run fwd_lanes2 DR_FWD_LANES=1 timeout 300 python bench.py $T
run infer timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 100 --warmup 10 --mode infer
run infer_lanes DR_EVAL_LANES=1 DR_GRAPHS=0 timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 100 --warmup 10 --mode infer
run infer_lanes_r1 DR_EVAL_LANES=1 DR_GRAPHS=0 timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 100 --warmup 10 --mode infer --replicas 1
run infer_r1 timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 100 --warmup 10 --mode infer --replicas 1
