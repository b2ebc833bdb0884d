#!/usr/bin/env bash
# GPU visit 3: conv micro-benchmark (tiles + ablations), gpu tests, bench train with per-layer detail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python tools/conv_bench.py > gpurun_out/conv_bench.md 2> gpurun_out/conv_bench.err; echo "rc=$?" >> gpurun_out/conv_bench.err
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 900 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline --detail gpurun_out/detail_train.md > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err; echo "bench rc=$?" >> gpurun_out/bench_train.err
timeout 600 python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline --detail gpurun_out/detail_infer.md > gpurun_out/bench_infer.json 2> gpurun_out/bench_infer.err
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -3; cut -c1-400 gpurun_out/bench_train.json; echo; cut -c1-300 gpurun_out/bench_infer.json; echo; cat gpurun_out/conv_bench.md | head -80
