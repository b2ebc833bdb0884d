#!/usr/bin/env bash
# round 5, visit 14: A/B of the 128-column x3 tile as 64-row workgroups of four waves (three independent workgroups per CU) against the
# eight-wave 128-row workgroups (DR_X3_VARIANT=8)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
Q="--steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > gpurun_out/v14_$name.json 2> gpurun_out/v14_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/v14_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)"; }
run w8_1 A=1
run r64_1 DR_X3_VARIANT=8
run w8_2 A=1
run r64_2 DR_X3_VARIANT=8
run w8_3 A=1
run r64_3 DR_X3_VARIANT=8
echo "== eight waves"; timeout 300 python tools/x3_bench.py 200 2>/dev/null | sed -n 1,11p | cut -c1-75
echo "== 64-row workgroups"; DR_X3_VARIANT=8 timeout 300 python tools/x3_bench.py 200 2>/dev/null | sed -n 3,11p | cut -c1-75
DR_X3_VARIANT=8 timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_trained_parity.py -q -m gpu -p no:cacheprovider > gpurun_out/v14_parity_r64.log 2>&1; echo "rc=$?" >> gpurun_out/v14_parity_r64.log
grep -v "start\]\|passed\]" gpurun_out/v14_parity_r64.log | tail -3
timeout 300 python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline --no-profile > gpurun_out/v14_infer.json 2>/dev/null; DR_X3_VARIANT=8 timeout 300 python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline --no-profile > gpurun_out/v14_infer_r64.json 2>/dev/null
python -c "
import json
for n in ('infer','infer_r64'):
    d=json.load(open('gpurun_out/v14_%s.json'%n)); print(n, round(d['value'],1), d['config']['single_replica'])"
