#!/usr/bin/env bash
# round 2, visit 4: look-back hand-off, second form (one acquire per workgroup, plain coefficient loads)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 200 python -m pytest tests/test_bn_layer.py tests/test_train_parity.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r02_pytest_gpu4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_pytest_gpu4.log
B="timeout 120 python bench.py --no-cpu-baseline --no-forward-vote --steps 60 --warmup 10"
DR_BN_LOOKBACK=1 $B > gpurun_out/ab_lb.json 2> gpurun_out/ab_lb.err
DR_BN_LOOKBACK=0 $B > gpurun_out/ab_nolb.json 2> gpurun_out/ab_nolb.err
DR_BN_LOOKBACK=1 $B > gpurun_out/ab_lb2.json 2> gpurun_out/ab_lb2.err
tail -4 gpurun_out/r02_pytest_gpu4.log
for f in lb nolb lb2; do python - <<PY
import json
try:
    d=json.load(open('gpurun_out/ab_$f.json')); k=d['roofline']['all_kernels']
    print('$f', round(d['value'],1), 'crops/s', round(d['ms_per_step'],3), 'ms |', ' '.join('%s=%.2f(%d)'%(n,v['ms_per_step'],v['launches']) for n,v in k.items() if v['ms_per_step']>0.2))
except Exception as e:
    print('$f', 'failed', e)
PY
done
