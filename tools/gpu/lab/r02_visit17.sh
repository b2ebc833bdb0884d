#!/usr/bin/env bash
# round 2, visit 17: loss kernel with one joint per workgroup row; workgroup cap of the grid-stride elementwise kernels
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 200 python -m pytest tests/test_train_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "single_stack or config3 or input_256 or trajectory" > gpurun_out/r02_pytest_gpu17.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_pytest_gpu17.log
B="timeout 120 python bench.py --no-cpu-baseline --no-forward-vote --steps 60 --warmup 10"
for m in 2048 512 1024 4096; do DR_ELT_GRID=$m $B > gpurun_out/ab_el$m.json 2> gpurun_out/ab_el$m.err; done
tail -3 gpurun_out/r02_pytest_gpu17.log
for m in 2048 512 1024 4096; do python - <<PY
import json
try:
    d=json.load(open('gpurun_out/ab_el$m.json')); k=d['roofline']['all_kernels']
    print('DR_ELT_GRID=$m', round(d['value'],1), 'crops/s', round(d['ms_per_step'],3), 'ms | loss', round(k['loss']['ms_per_step'],3), 'eltwise', round(k['eltwise_bwd']['ms_per_step'],3), 'pool', round(k['maxpool']['ms_per_step'],3), 'upadd', round(k['upsample_add']['ms_per_step'],3), 'copy', round(k['copy_channels']['ms_per_step'],3))
except Exception as e:
    print('$m', 'failed', e, open('gpurun_out/ab_el$m.err').read()[-300:])
PY
done
