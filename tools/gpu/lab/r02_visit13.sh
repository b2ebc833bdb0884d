#!/usr/bin/env bash
# round 2, visit 13: narrow-output K-split tiles (64x96, 64x160): micro-benchmark, parity on the GPU, step A/B, config 5
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 200 python tools/conv_narrow_bench.py > gpurun_out/conv_narrow.md 2> gpurun_out/conv_narrow.err
timeout 300 python -m pytest tests/test_forward_parity.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r02_pytest_gpu13.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_pytest_gpu13.log
B="timeout 120 python bench.py --no-cpu-baseline --no-forward-vote --steps 60 --warmup 10"
DR_CONV_NARROW=0 $B > gpurun_out/ab_n0.json 2> gpurun_out/ab_n0.err
DR_CONV_NARROW=1 $B > gpurun_out/ab_n1.json 2> gpurun_out/ab_n1.err
C5="--num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu --no-cpu-baseline --no-forward-vote --steps 20 --warmup 5"
DR_CONV_NARROW=0 timeout 200 python bench.py $C5 > gpurun_out/ab_c5n0.json 2> gpurun_out/ab_c5n0.err
DR_CONV_NARROW=1 timeout 200 python bench.py $C5 > gpurun_out/ab_c5n1.json 2> gpurun_out/ab_c5n1.err
DR_CONV_NARROW=1 timeout 200 python bench.py $C5 --precision bf16 > gpurun_out/ab_c5n1b.json 2> gpurun_out/ab_c5n1b.err
cat gpurun_out/conv_narrow.md; tail -3 gpurun_out/r02_pytest_gpu13.log
for m in n0 n1 c5n0 c5n1 c5n1b; do python - <<PY
import json
try:
    d=json.load(open('gpurun_out/ab_$m.json')); k=d['roofline']['all_kernels']
    print('$m', round(d['value'],1), 'crops/s', round(d['ms_per_step'],3), 'ms', d['dtype'], '|', ' '.join('%s=%.2f(%d,%.0fTF)'%(n,v['ms_per_step'],v['launches'],v['tflops'] or 0) for n,v in k.items() if n.startswith('conv_igemm')))
except Exception as e:
    print('$m', 'failed', e, open('gpurun_out/ab_$m.err').read()[-300:])
PY
done
