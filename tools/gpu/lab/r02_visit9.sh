#!/usr/bin/env bash
# round 2, visit 9: side-stream weight gradients restricted to a subset of the CUs
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
B="timeout 120 python bench.py --no-cpu-baseline --no-forward-vote --no-profile --steps 80 --warmup 10"
for m in 0 224 192 160 128 96; do DR_WGRAD_CUS=$m $B > gpurun_out/ab_cu$m.json 2> gpurun_out/ab_cu$m.err; done
for m in 0 224 192 160 128 96; do python - <<PY
import json
try:
    d=json.load(open('gpurun_out/ab_cu$m.json'))
    print('DR_WGRAD_CUS=$m', round(d['value'],1), 'crops/s', round(d['ms_per_step'],3), 'ms')
except Exception as e:
    print('$m', 'failed', e, open('gpurun_out/ab_cu$m.err').read()[-300:])
PY
done
