#!/usr/bin/env bash
# GPU visit: bf16 matrix-core conv path -- parity tests, per-tile micro-benchmark, inference bench in both precisions
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_forward_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "bf16" > gpurun_out/pytest_bf16.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_bf16.log
tail -4 gpurun_out/pytest_bf16.log
timeout 300 python tools/conv_bench_bf16.py > gpurun_out/conv_bench_bf16.md 2> gpurun_out/conv_bench_bf16.err
cat gpurun_out/conv_bench_bf16.md
for prec in f32 bf16; do
  timeout 300 python bench.py --mode infer --precision $prec --steps 40 --warmup 10 --no-cpu-baseline --detail gpurun_out/detail_infer_$prec.md > gpurun_out/bench_infer_$prec.json 2> gpurun_out/bench_infer_$prec.err
  cut -c1-260 gpurun_out/bench_infer_$prec.json; echo
done
