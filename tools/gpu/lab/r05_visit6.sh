#!/usr/bin/env bash
# round 5, visit 6: A/B of the x3 rules (grid threshold, weight-gradient tile rule) on the training step; config 5 fp32 / MSRA lines
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
Q="--steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > gpurun_out/v6_$name.json 2> gpurun_out/v6_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/v6_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)"; }
run base A=1
run wgs256 DR_X3_MIN_WGS=256
run wgs384 DR_X3_MIN_WGS=384
run t128 DR_WG_X3_T128=1
run t128_wgs256 DR_WG_X3_T128=1 DR_X3_MIN_WGS=256
run base2 A=1
run oneacc DR_X3_VARIANT=1
timeout 300 python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline --no-profile > gpurun_out/v6_infer.json 2> gpurun_out/v6_infer.err; python -c "
import json; d=json.load(open('gpurun_out/v6_infer.json')); print('infer', round(d['value'],1), d['config']['single_replica'])"
DR_X3_MIN_WGS=256 timeout 300 python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline --no-profile > gpurun_out/v6_infer256.json 2> gpurun_out/v6_infer256.err; python -c "
import json; d=json.load(open('gpurun_out/v6_infer256.json')); print('infer wgs256', round(d['value'],1), d['config']['single_replica'])"
timeout 400 python bench.py --num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu --no-cpu-baseline --steps 10 --warmup 5 --no-forward-vote --no-profile > gpurun_out/v6_c5_f32.json 2> gpurun_out/v6_c5_f32.err; python -c "
import json; d=json.load(open('gpurun_out/v6_c5_f32.json')); print('config5 fp32', round(d['value'],1), round(d['ms_per_step'],2))"
