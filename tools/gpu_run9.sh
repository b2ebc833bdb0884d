#!/usr/bin/env bash
# GPU visit: full gpu test suite, smoke, bench train+infer with per-layer detail (no rocprof passes)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 --detail gpurun_out/detail_train.md > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err; echo "bench rc=$?" >> gpurun_out/bench_train.err
timeout 600 python bench.py --mode infer --steps 20 --warmup 5 --detail gpurun_out/detail_infer.md > gpurun_out/bench_infer.json 2> gpurun_out/bench_infer.err
DR_SINGLE_STREAM=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile > gpurun_out/bench_train_1s.json 2> gpurun_out/bench_train_1s.err
DR_SINGLE_STREAM=1 timeout 600 python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline --no-profile > gpurun_out/bench_infer_1s.json 2> gpurun_out/bench_infer_1s.err
cut -c1-200 gpurun_out/bench_train_1s.json; echo; cut -c1-200 gpurun_out/bench_infer_1s.json; echo
grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -5; tail -2 gpurun_out/smoke.log; cut -c1-700 gpurun_out/bench_train.json; echo; cut -c1-700 gpurun_out/bench_infer.json; echo
