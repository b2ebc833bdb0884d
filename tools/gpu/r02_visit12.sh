#!/usr/bin/env bash
# round 2, visit 12: slab folds on the side stream; bf16 with the side stream
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 200 python -m pytest tests/test_train_parity.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r02_pytest_gpu12.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_pytest_gpu12.log
B="timeout 120 python bench.py --no-cpu-baseline --no-forward-vote --no-profile --steps 80 --warmup 10"
$B > gpurun_out/ab_f1.json 2> gpurun_out/ab_f1.err
$B > gpurun_out/ab_f2.json 2> gpurun_out/ab_f2.err
$B --precision bf16 > gpurun_out/ab_b1.json 2> gpurun_out/ab_b1.err
DR_WGRAD_STREAM=0 $B --precision bf16 > gpurun_out/ab_b0.json 2> gpurun_out/ab_b0.err
tail -3 gpurun_out/r02_pytest_gpu12.log
for m in f1 f2 b1 b0; do python - <<PY
import json
try:
    d=json.load(open('gpurun_out/ab_$m.json'))
    print('$m', round(d['value'],1), 'crops/s', round(d['ms_per_step'],3), 'ms', d['dtype'])
except Exception as e:
    print('$m', 'failed', e, open('gpurun_out/ab_$m.err').read()[-300:])
PY
done
