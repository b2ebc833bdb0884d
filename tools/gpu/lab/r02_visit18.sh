#!/usr/bin/env bash
# round 2, visit 18: workgroup cap of the BatchReNorm backward reduce pass
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
B="timeout 120 python bench.py --no-cpu-baseline --no-forward-vote --steps 60 --warmup 10"
for m in 256 128 512 1024; do DR_BN_RED_GRID=$m $B > gpurun_out/ab_rd$m.json 2> gpurun_out/ab_rd$m.err; done
for m in 256 128 512 1024; do python - <<PY
import json
try:
    d=json.load(open('gpurun_out/ab_rd$m.json')); k=d['roofline']['all_kernels']
    print('DR_BN_RED_GRID=$m', round(d['value'],1), 'crops/s', round(d['ms_per_step'],3), 'ms | batch_renorm', round(k['batch_renorm']['ms_per_step'],3))
except Exception as e:
    print('$m', 'failed', e, open('gpurun_out/ab_rd$m.err').read()[-300:])
PY
done
