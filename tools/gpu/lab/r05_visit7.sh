#!/usr/bin/env bash
# round 5, visit 7: the three-stage ring of conv_x3.h against the two-stage kernel: tests, micro-benchmark, training / inference step
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 300 python -m pytest tests/test_forward_parity.py -q -m gpu -k "x3" -p no:cacheprovider > gpurun_out/v7_x3_tests.log 2>&1; echo "rc=$?" >> gpurun_out/v7_x3_tests.log
timeout 600 python tools/x3_bench.py 200 > gpurun_out/v7_x3_bench_b200.md 2> gpurun_out/v7_x3_bench_b200.err
Q="--steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > gpurun_out/v7_$name.json 2> gpurun_out/v7_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/v7_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)"; }
run ring A=1
run twostage DR_X3_VARIANT=2
run ring2 A=1
run twostage2 DR_X3_VARIANT=2
timeout 300 python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline --no-profile > gpurun_out/v7_infer.json 2> gpurun_out/v7_infer.err; python -c "
import json; d=json.load(open('gpurun_out/v7_infer.json')); print('infer ring', round(d['value'],1), d['config']['single_replica'])"
DR_X3_VARIANT=2 timeout 300 python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline --no-profile > gpurun_out/v7_infer2.json 2> gpurun_out/v7_infer2.err; python -c "
import json; d=json.load(open('gpurun_out/v7_infer2.json')); print('infer two-stage', round(d['value'],1), d['config']['single_replica'])"
timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_trained_parity.py -q -m gpu -p no:cacheprovider > gpurun_out/v7_parity.log 2>&1; echo "rc=$?" >> gpurun_out/v7_parity.log
tail -3 gpurun_out/v7_x3_tests.log; cat gpurun_out/v7_x3_bench_b200.md; tail -3 gpurun_out/v7_x3_bench_b200.err; grep -v "start\]\|passed\]" gpurun_out/v7_parity.log | tail -4
