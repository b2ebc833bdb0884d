#!/usr/bin/env bash
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
Q="--no-cpu-baseline --no-profile --no-forward-vote --steps 50 --warmup 10"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > $G/v17_$name.json 2> $G/v17_$name.err; python -c "
import json;d=json.load(open('$G/v17_$name.json'));print('$name',round(d['value'],1),round(d['ms_per_step'],3))" 2>/dev/null || { echo "$name FAILED"; tail -5 $G/v17_$name.err; }; }
run base DR_PIPELINE=2
run wgnormal DR_WG_PRIO=0
run wgnormal_q6 DR_WG_PRIO=0 GPU_MAX_HW_QUEUES=6
run wgnormal_q8 DR_WG_PRIO=0 GPU_MAX_HW_QUEUES=8
run q3 GPU_MAX_HW_QUEUES=3
run q5 GPU_MAX_HW_QUEUES=5
run q6 GPU_MAX_HW_QUEUES=6
run base2 DR_PIPELINE=2
