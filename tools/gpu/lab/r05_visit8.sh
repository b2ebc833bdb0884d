#!/usr/bin/env bash
# round 5, visit 8: conv_x3 with eight waves per workgroup (default) against four waves and the register-staged ring
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 300 python -m pytest tests/test_forward_parity.py -q -m gpu -k "x3" -p no:cacheprovider > gpurun_out/v8_x3_tests.log 2>&1; echo "rc=$?" >> gpurun_out/v8_x3_tests.log
timeout 600 python tools/x3_bench.py 200 > gpurun_out/v8_x3_bench_b200.md 2> gpurun_out/v8_x3_bench_b200.err
timeout 300 python tools/x3_bench.py 40 > gpurun_out/v8_x3_bench_b40.md 2> gpurun_out/v8_x3_bench_b40.err
Q="--steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > gpurun_out/v8_$name.json 2> gpurun_out/v8_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/v8_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)"; }
run w8 A=1
run w4 DR_X3_VARIANT=4
run ring DR_X3_VARIANT=2
run w8b A=1
run w4b DR_X3_VARIANT=4
timeout 300 python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline --no-profile > gpurun_out/v8_infer.json 2> gpurun_out/v8_infer.err; python -c "
import json; d=json.load(open('gpurun_out/v8_infer.json')); print('infer w8', round(d['value'],1), d['config']['single_replica'])"
DR_X3_VARIANT=4 timeout 300 python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline --no-profile > gpurun_out/v8_infer4.json 2> gpurun_out/v8_infer4.err; python -c "
import json; d=json.load(open('gpurun_out/v8_infer4.json')); print('infer w4', round(d['value'],1), d['config']['single_replica'])"
tail -3 gpurun_out/v8_x3_tests.log; cat gpurun_out/v8_x3_bench_b200.md; tail -3 gpurun_out/v8_x3_bench_b200.err;  cat gpurun_out/v8_x3_bench_b40.md | head -12
