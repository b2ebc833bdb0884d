#!/usr/bin/env bash
# first GPU visit: gpu tests, smoke, bench(infer), rocprof kernel stats
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rocminfo | grep -E "Marketing|gfx" | head -4 > gpurun_out/device.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 600 python bench.py --mode infer --steps 20 --warmup 5 > gpurun_out/bench_infer.json 2> gpurun_out/bench_infer.err; echo "bench rc=$?" >> gpurun_out/bench_infer.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_infer -o infer -- python $R/bench.py --mode infer --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $R/gpurun_out/rocprof_infer.log 2>&1; echo "rocprof rc=$?" >> $R/gpurun_out/rocprof_infer.log
cd $R; ls -R gpurun_out | head -50
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log | tail -3; cat gpurun_out/bench_infer.json | cut -c1-1500
