#!/usr/bin/env bash
# GPU visit 2: all gpu tests, smoke, bench train + infer with per-layer detail, rocprof kernel stats (train)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python bench.py --mode train --steps 20 --warmup 5 --detail gpurun_out/detail_train.md > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err; echo "bench rc=$?" >> gpurun_out/bench_train.err
timeout 600 python bench.py --mode infer --steps 20 --warmup 5 --detail gpurun_out/detail_infer.md > gpurun_out/bench_infer.json 2> gpurun_out/bench_infer.err; echo "bench rc=$?" >> gpurun_out/bench_infer.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train -o train -- python $R/bench.py --mode train --steps 10 --warmup 5 --no-cpu-baseline --no-profile > $R/gpurun_out/rocprof_train.log 2>&1; echo "rocprof rc=$?" >> $R/gpurun_out/rocprof_train.log
cd $R
grep -E "passed|failed|grad error" gpurun_out/pytest_gpu.log | tail -8; tail -2 gpurun_out/smoke.log; cut -c1-700 gpurun_out/bench_train.json; tail -3 gpurun_out/bench_train.err
