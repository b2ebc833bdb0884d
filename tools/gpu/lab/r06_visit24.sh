#!/usr/bin/env bash
# round 6, visit 24: BatchReNorm streaming passes with non-temporal stores (nt1), loads (nt2), both (nt3): build variants, A/B on the step
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
Q="--steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > gpurun_out/r06v24_$name.json 2> gpurun_out/r06v24_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/r06v24_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)"; }
for i in 1 2; do
run base_$i A=1
run nt1_$i DR_LIB_VARIANT=nt1
run nt2_$i DR_LIB_VARIANT=nt2
run nt3_$i DR_LIB_VARIANT=nt3
done
