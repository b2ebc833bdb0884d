#!/usr/bin/env bash
# round 6, visit 25: non-temporal hints beyond the BatchReNorm passes (now default): conv epilogue stores (ep), fold loads + slab stores (fold), all
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
Q="--steps 10 --warmup 5 --no-cpu-baseline --no-profile"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > gpurun_out/r06v25_$name.json 2> gpurun_out/r06v25_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/r06v25_$name.json')); fv=d.get('forward_vote') or {}; print('$name', round(d['value'],1), round(d['ms_per_step'],3), 'single', round((fv.get('single_replica') or {}).get('value',0),1))
except Exception as e: print('$name failed', e)"; }
for i in 1 2; do
run base_$i A=1
run ep_$i DR_LIB_VARIANT=ep
run fold_$i DR_LIB_VARIANT=fold
run all_$i DR_LIB_VARIANT=all
done
