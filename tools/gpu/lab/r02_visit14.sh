#!/usr/bin/env bash
# round 2, visit 14: bf16 storage of dRaw on the bf16 matrix-core path
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 300 python -m pytest tests/test_train_parity.py tests/test_gpu_configs.py -m gpu -q --tb=short -p no:cacheprovider -k "bf16 or config5" > gpurun_out/r02_pytest_gpu14.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_pytest_gpu14.log
B="timeout 200 python bench.py --no-cpu-baseline --no-forward-vote --precision bf16"
DR_BF16_DRAW=0 $B --steps 60 --warmup 10 > gpurun_out/ab_d0.json 2> gpurun_out/ab_d0.err
DR_BF16_DRAW=1 $B --steps 60 --warmup 10 > gpurun_out/ab_d1.json 2> gpurun_out/ab_d1.err
C5="--num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu --steps 20 --warmup 5"
DR_BF16_DRAW=0 $B $C5 > gpurun_out/ab_c5d0.json 2> gpurun_out/ab_c5d0.err
DR_BF16_DRAW=1 $B $C5 > gpurun_out/ab_c5d1.json 2> gpurun_out/ab_c5d1.err
tail -3 gpurun_out/r02_pytest_gpu14.log
for m in d0 d1 c5d0 c5d1; do python - <<PY
import json
try:
    d=json.load(open('gpurun_out/ab_$m.json')); k=d['roofline']['all_kernels']
    print('$m', round(d['value'],1), 'crops/s', round(d['ms_per_step'],3), 'ms', d['dtype'], '|', ' '.join('%s=%.2f'%(n,v['ms_per_step']) for n,v in k.items() if v['ms_per_step']>1.0 or n in ('conv_wgrad','batch_renorm')))
except Exception as e:
    print('$m', 'failed', e, open('gpurun_out/ab_$m.err').read()[-300:])
PY
done
