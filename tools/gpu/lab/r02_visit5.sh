#!/usr/bin/env bash
# round 2, visit 5: LDS-DMA refill with separate LDS stage objects (no vmcnt(0) in front of the fragment reads): micro + step A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
DR_CONV_GLDS=0 timeout 120 python tools/conv_ab.py > gpurun_out/conv_ab_glds0.md 2>&1
DR_CONV_GLDS=1 timeout 120 python tools/conv_ab.py > gpurun_out/conv_ab_glds1.md 2>&1
DR_CONV_GLDS=1 timeout 200 python -m pytest tests/test_forward_parity.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r02_pytest_glds.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_pytest_glds.log
B="timeout 120 python bench.py --no-cpu-baseline --no-forward-vote --steps 60 --warmup 10"
DR_CONV_GLDS=0 $B > gpurun_out/ab_g0.json 2> gpurun_out/ab_g0.err
DR_CONV_GLDS=1 $B > gpurun_out/ab_g1.json 2> gpurun_out/ab_g1.err
DR_CONV_GLDS=0 $B --mode infer > gpurun_out/ab_g0i.json 2> gpurun_out/ab_g0i.err
DR_CONV_GLDS=1 $B --mode infer > gpurun_out/ab_g1i.json 2> gpurun_out/ab_g1i.err
paste -d' ' gpurun_out/conv_ab_glds0.md gpurun_out/conv_ab_glds1.md | cut -c1-200
tail -3 gpurun_out/r02_pytest_glds.log
for f in g0 g1 g0i g1i; do python - <<PY
import json
try:
    d=json.load(open('gpurun_out/ab_$f.json')); k=d['roofline']['all_kernels']
    print('$f', round(d['value'],1), 'crops/s', round(d['ms_per_step'],3), 'ms |', ' '.join('%s=%.2f(%d)'%(n,v['ms_per_step'],v['launches']) for n,v in k.items() if v['ms_per_step']>0.2))
except Exception as e:
    print('$f', 'failed', e)
PY
done
