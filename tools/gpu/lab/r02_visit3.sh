#!/usr/bin/env bash
# round 2, visit 3: look-back hand-off of the BatchReNorm coefficients: gpu tests, A/B against finalize launches
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 300 python -m pytest tests/test_bn_layer.py tests/test_train_parity.py tests/test_gpu_configs.py::test_config5_s4_f256_in256_train_b1 -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r02_pytest_gpu3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_pytest_gpu3.log
B="timeout 120 python bench.py --no-cpu-baseline --no-forward-vote --steps 60 --warmup 10"
DR_BN_LOOKBACK=0 $B > gpurun_out/ab_nolb.json 2> gpurun_out/ab_nolb.err
DR_BN_LOOKBACK=1 $B > gpurun_out/ab_lb.json 2> gpurun_out/ab_lb.err
DR_BN_LOOKBACK=0 $B > gpurun_out/ab_nolb2.json 2> gpurun_out/ab_nolb2.err
DR_BN_LOOKBACK=1 $B > gpurun_out/ab_lb2.json 2> gpurun_out/ab_lb2.err
tail -8 gpurun_out/r02_pytest_gpu3.log
for f in nolb lb nolb2 lb2; do python - <<PY
import json
try:
    d=json.load(open('gpurun_out/ab_$f.json')); k=d['roofline']['all_kernels']
    print('$f', round(d['value'],1), 'crops/s', round(d['ms_per_step'],3), 'ms |', ' '.join('%s=%.2f(%d)'%(n,v['ms_per_step'],v['launches']) for n,v in k.items() if v['ms_per_step']>0.2))
except Exception as e:
    print('$f', 'failed', e)
PY
done
