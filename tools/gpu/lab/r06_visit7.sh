#!/usr/bin/env bash
# round 6, visit 7: the halo kernel on every 3x3 tile width and from one workgroup per CU on -- training step, inference, GPU parity suites
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
Q="--steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > gpurun_out/r06v7_$name.json 2> gpurun_out/r06v7_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/r06v7_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)"; }
run halo_1 A=1
run nohalo_1 DR_X3_HALO=0
run halo_2 A=1
run nohalo_2 DR_X3_HALO=0
for n in infer infer_nohalo; do
  if [ $n = infer ]; then E="A=1"; else E="DR_X3_HALO=0"; fi
  env $E timeout 300 python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline --no-profile > gpurun_out/r06v7_$n.json 2>/dev/null
  python -c "
import json
d=json.load(open('gpurun_out/r06v7_$n.json')); print('$n', round(d['value'],1), d['config'].get('single_replica'))"
done
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_trained_parity.py tests/test_train_parity.py tests/test_bench_shapes.py tests/test_forward_parity.py -q -m gpu -p no:cacheprovider -x > gpurun_out/r06v7_parity.log 2>&1; echo "rc=$?" >> gpurun_out/r06v7_parity.log
grep -v "start\]\|passed\]" gpurun_out/r06v7_parity.log | tail -8
