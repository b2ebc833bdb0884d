#!/usr/bin/env bash
# round 2, visit 7: side-stream weight gradients, release policy sweep
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
B="timeout 120 python bench.py --no-cpu-baseline --no-forward-vote --no-profile --steps 80 --warmup 10"
for m in 1 2 3 5 9; do DR_WGRAD_STREAM=$m $B > gpurun_out/ab_ws$m.json 2> gpurun_out/ab_ws$m.err; done
for m in 1 2 3 5 9; do python - <<PY
import json
try:
    d=json.load(open('gpurun_out/ab_ws$m.json'))
    print('DR_WGRAD_STREAM=$m', round(d['value'],1), 'crops/s', round(d['ms_per_step'],3), 'ms')
except Exception as e:
    print('$m', 'failed', e, open('gpurun_out/ab_ws$m.err').read()[-300:])
PY
done
