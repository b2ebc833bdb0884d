#!/usr/bin/env bash
# round 3, visit 32: the window as TWO passes in flight on the two micro-step slots (3 + 2 micro-batches) against one pass of 5
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
T="--no-cpu-baseline --no-profile --no-forward-vote --steps 100 --warmup 10"
run() { name=$1; shift; env "$@" > $G/v32_$name.json 2> $G/v32_$name.err; python -c "
import json;d=json.load(open('$G/v32_$name.json'));print('$name',round(d['value'],1),round(d['ms_per_step'],3),d['config'].get('micro_steps_per_pass'),d['config'].get('micro_steps_in_flight'))" 2>/dev/null || { echo "$name FAILED"; tail -5 $G/v32_$name.err; }; }
run one_pass timeout 300 python bench.py $T
run split_3_2 DR_PIPELINE=2 timeout 300 python bench.py $T
run depth2_nosplit DR_PIPELINE=2 DR_WINDOW_SPLIT=0 timeout 300 python bench.py $T
run split_bf16 DR_PIPELINE=2 timeout 300 python bench.py $T --precision bf16
run one_pass_bf16 timeout 300 python bench.py $T --precision bf16
run split_sub8 DR_PIPELINE=2 timeout 300 python bench.py $T --sub_batch 8 --steps 96 --warmup 16
run one_pass_sub8 timeout 300 python bench.py $T --sub_batch 8 --steps 96 --warmup 16
