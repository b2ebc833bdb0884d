#!/usr/bin/env bash
# round 6, visit 18: conv epilogue statistics -- fp32 sums per batch of rows joined to the fp64 partial once (and skipped in eval mode) against
# three fp64 instructions per element (lib/variants/dpstats = the build before)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_forward_parity.py tests/test_bn_layer.py tests/test_train_parity.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
Q="--steps 10 --warmup 5 --no-cpu-baseline --no-profile"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > gpurun_out/r06v18_$name.json 2> gpurun_out/r06v18_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/r06v18_$name.json')); fv=d.get('forward_vote') or {}; print('$name', round(d['value'],1), round(d['ms_per_step'],3), 'fwd+vote', round(fv.get('value',0),1), 'single', round((fv.get('single_replica') or {}).get('value',0),1))
except Exception as e: print('$name failed', e)"; }
run new_1 A=1
run old_1 DR_LIB_VARIANT=dpstats
run new_2 A=1
run old_2 DR_LIB_VARIANT=dpstats
run new_3 A=1
run old_3 DR_LIB_VARIANT=dpstats
