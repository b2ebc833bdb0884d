#!/usr/bin/env bash
# GPU visit: bf16 training path -- parity tests, training bench in both precisions with per-layer detail
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_train_parity.py tests/test_forward_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "bf16" > gpurun_out/pytest_bf16_train.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_bf16_train.log
tail -12 gpurun_out/pytest_bf16_train.log
for prec in bf16 f32; do
  timeout 300 python bench.py --mode train --precision $prec --steps 20 --warmup 5 --no-cpu-baseline --detail gpurun_out/detail_train_$prec.md > gpurun_out/bench_train_$prec.json 2> gpurun_out/bench_train_$prec.err
  cut -c1-260 gpurun_out/bench_train_$prec.json; echo; tail -2 gpurun_out/bench_train_$prec.err
done
