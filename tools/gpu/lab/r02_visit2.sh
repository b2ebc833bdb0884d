#!/usr/bin/env bash
# round 2, visit 2: re-run the changed gpu tests, A/B of the backward-sweep changes, the default bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 420 python -m pytest tests/test_gpu_configs.py tests/test_bn_layer.py tests/test_train_parity.py -m gpu -q --tb=short -p no:cacheprovider --durations=8 > gpurun_out/r02_pytest_gpu2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_pytest_gpu2.log
B="timeout 120 python bench.py --no-cpu-baseline --no-forward-vote --steps 60 --warmup 10"
DR_GROUP_WGRAD=0 DR_FUSE_ACT=0 DR_FUSE_LAST=0 $B > gpurun_out/ab_base.json 2> gpurun_out/ab_base.err
DR_GROUP_WGRAD=1 DR_FUSE_ACT=0 DR_FUSE_LAST=0 $B > gpurun_out/ab_group.json 2> gpurun_out/ab_group.err
DR_GROUP_WGRAD=0 DR_FUSE_ACT=1 DR_FUSE_LAST=0 $B > gpurun_out/ab_act.json 2> gpurun_out/ab_act.err
DR_GROUP_WGRAD=0 DR_FUSE_ACT=0 DR_FUSE_LAST=1 $B > gpurun_out/ab_last.json 2> gpurun_out/ab_last.err
timeout 240 python bench.py --detail gpurun_out/r02_detail_train.md > gpurun_out/r02_bench_train.json 2> gpurun_out/r02_bench_train.err; echo "bench rc=$?" >> gpurun_out/r02_bench_train.err
tail -12 gpurun_out/r02_pytest_gpu2.log
for f in base group act last; do python - <<PY
import json
try:
    d=json.load(open('gpurun_out/ab_$f.json')); k=d['roofline']['all_kernels']
    print('$f', round(d['value'],1), 'crops/s', round(d['ms_per_step'],3), 'ms |', ' '.join('%s=%.2f'%(n,v['ms_per_step']) for n,v in k.items() if v['ms_per_step']>0.2))
except Exception as e:
    print('$f', 'failed', e)
PY
done
cut -c1-600 gpurun_out/r02_bench_train.json; tail -2 gpurun_out/r02_bench_train.err
