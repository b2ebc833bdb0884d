#!/usr/bin/env bash
# round 6, visit 22: the halo rule ahead of the 128-column rule (3x3 128 -> 128 of a 40-crop batch on conv_x3h_kernel), DR_X3_MIN_WGS at B = 40
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_forward_parity.py tests/test_gpu_fullsize.py tests/test_trained_parity.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
Q="--mode infer --replicas 1 --merge 1 --steps 100 --warmup 10 --no-cpu-baseline --no-profile"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > gpurun_out/r06v22_$name.json 2> gpurun_out/r06v22_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/r06v22_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)"; }
run new_1 A=1
run old_1 DR_LIB_VARIANT=dpstats
run wgs256_1 DR_X3_MIN_WGS=256
run wgs320_1 DR_X3_MIN_WGS=320
run new_2 A=1
run old_2 DR_LIB_VARIANT=dpstats
run wgs256_2 DR_X3_MIN_WGS=256
run wgs320_2 DR_X3_MIN_WGS=320
Q="--mode infer --steps 50 --warmup 10 --no-cpu-baseline --no-profile"
run pool_new A=1
run pool_old DR_LIB_VARIANT=dpstats
