#!/usr/bin/env bash
# round 6, visit 31: two K-tiles of pixels REALLY in flight (DR_X3_PF=1: global_load_dwordx4 as inline asm, waits tied to the registers)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
DR_X3_PF=1 timeout 600 python -m pytest tests/test_forward_parity.py -q -m gpu -k "conv_x3" -p no:cacheprovider -x 2>&1 | tail -2
for m in 0 1 0 1; do echo "DR_X3_PF=$m"; DR_X3_PF=$m timeout 300 python tools/x3_bn256_bench.py 200 2>/dev/null | sed -n 3,11p; done | tee gpurun_out/r06v31_pf.md
DR_X3_PF=1 timeout 120 python tools/x3_intercept_bench.py 200 2>/dev/null | tail -3
Q="--steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > gpurun_out/r06v31_$name.json 2> gpurun_out/r06v31_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/r06v31_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)"; }
