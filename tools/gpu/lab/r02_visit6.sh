#!/usr/bin/env bash
# round 2, visit 6: full-resolution weight gradients on a low-priority side stream, beside the hourglass chains
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 200 python -m pytest tests/test_train_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "grouped or config3 or single_stack" > gpurun_out/r02_pytest_gpu6.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_pytest_gpu6.log
B="timeout 120 python bench.py --no-cpu-baseline --no-forward-vote --no-profile --steps 80 --warmup 10"
DR_WGRAD_STREAM=0 $B > gpurun_out/ab_s0.json 2> gpurun_out/ab_s0.err
DR_WGRAD_STREAM=1 $B > gpurun_out/ab_s1.json 2> gpurun_out/ab_s1.err
DR_WGRAD_STREAM=0 $B > gpurun_out/ab_s0b.json 2> gpurun_out/ab_s0b.err
DR_WGRAD_STREAM=1 $B > gpurun_out/ab_s1b.json 2> gpurun_out/ab_s1b.err
tail -3 gpurun_out/r02_pytest_gpu6.log
for f in s0 s1 s0b s1b; do python - <<PY
import json
try:
    d=json.load(open('gpurun_out/ab_$f.json'))
    print('$f', round(d['value'],1), 'crops/s', round(d['ms_per_step'],3), 'ms')
except Exception as e:
    print('$f', 'failed', e, open('gpurun_out/ab_$f.err').read()[-300:])
PY
done
