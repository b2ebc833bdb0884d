#!/usr/bin/env bash
# round 6, visit 27: rows in flight per thread of the BatchReNorm passes (DR_BN_ROWS, build variants) with the non-temporal hints on
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
Q="--steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > gpurun_out/r06v27_$name.json 2> gpurun_out/r06v27_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/r06v27_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)"; }
for i in 1 2; do
run base_$i A=1
run rows2_$i DR_LIB_VARIANT=rows2
run rows6_$i DR_LIB_VARIANT=rows6
run rows8_$i DR_LIB_VARIANT=rows8
run grid1024_$i DR_BN_GRID=1024
run grid2048_$i DR_BN_GRID=2048
done
