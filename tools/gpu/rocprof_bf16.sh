#!/usr/bin/env bash
# GPU visit: rocprofv3 kernel stats of the bf16 matrix-core runs
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train_bf16 -o train -- python $R/bench.py --precision bf16 --steps 10 --warmup 5 --no-cpu-baseline --no-profile > $R/gpurun_out/rocprof_train_bf16.log 2>&1; echo "rc=$?"
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_infer_bf16 -o infer -- python $R/bench.py --mode infer --precision bf16 --steps 10 --warmup 5 --no-cpu-baseline --no-profile > $R/gpurun_out/rocprof_infer_bf16.log 2>&1; echo "rc=$?"
ls -la $R/gpurun_out/prof_train_bf16 $R/gpurun_out/prof_infer_bf16
