#!/usr/bin/env bash
# round 6, visit 23: one engine at B = 40, forward(eval) + vote: hipGraph replay and executor lanes, A/B
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
Q="--mode infer --replicas 1 --merge 1 --steps 100 --warmup 10 --no-cpu-baseline --no-profile"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > gpurun_out/r06v23_$name.json 2> gpurun_out/r06v23_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/r06v23_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)"; tail -2 gpurun_out/r06v23_$name.err; }
run base_1 A=1
run graphs_1 DR_GRAPHS=1
run lanes_1 DR_MULTI_STREAM=1
run both_1 DR_GRAPHS=1 DR_MULTI_STREAM=1
run base_2 A=1
run graphs_2 DR_GRAPHS=1
run lanes_2 DR_MULTI_STREAM=1
run both_2 DR_GRAPHS=1 DR_MULTI_STREAM=1
run nofuse DR_FUSION=0
