#!/usr/bin/env bash
# round 2, visit 10: side-stream weight gradients cut to ~N workgroups per launch
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
B="timeout 120 python bench.py --no-cpu-baseline --no-forward-vote --no-profile --steps 80 --warmup 10"
for m in 0 256 384 512 768; do DR_WGRAD_SIDE_WGS=$m $B > gpurun_out/ab_sw$m.json 2> gpurun_out/ab_sw$m.err; done
for m in 0 256 384 512 768; do python - <<PY
import json
try:
    d=json.load(open('gpurun_out/ab_sw$m.json'))
    print('DR_WGRAD_SIDE_WGS=$m', round(d['value'],1), 'crops/s', round(d['ms_per_step'],3), 'ms')
except Exception as e:
    print('$m', 'failed', e, open('gpurun_out/ab_sw$m.err').read()[-300:])
PY
done
