#!/usr/bin/env bash
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
Q="--no-cpu-baseline --no-profile --no-forward-vote --steps 40 --warmup 10"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > $G/v18_$name.json 2> $G/v18_$name.err; python -c "
import json;d=json.load(open('$G/v18_$name.json'));print('$name',round(d['value'],1),round(d['ms_per_step'],3))" 2>/dev/null || { echo "$name FAILED"; tail -5 $G/v18_$name.err; }; }
run d1_a DR_PIPELINE=1
run d1_b DR_PIPELINE=1
run d1_graphs DR_PIPELINE=1 DR_GRAPHS=1
run d1_nows DR_PIPELINE=1 DR_WGRAD_STREAM=0
run d1_wgnormal DR_PIPELINE=1 DR_WG_PRIO=0
run d2_a DR_PIPELINE=2
run d2_graphs DR_PIPELINE=2 DR_GRAPHS=1
run d1_c DR_PIPELINE=1
