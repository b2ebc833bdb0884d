#!/usr/bin/env bash
# GPU visit: conv microbench, gpu tests, 1-rank RCCL all-reduce path, bench + rocprof (train)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python tools/conv_bench.py > gpurun_out/conv_bench.md 2> gpurun_out/conv_bench.err
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
DR_FORCE_ALLREDUCE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 5 --no-cpu-baseline --no-profile > gpurun_out/bench_rccl1.json 2> gpurun_out/bench_rccl1.err; echo "rc=$?" >> gpurun_out/bench_rccl1.err
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --detail gpurun_out/detail_train.md > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err
timeout 600 python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline --detail gpurun_out/detail_infer.md > gpurun_out/bench_infer.json 2> gpurun_out/bench_infer.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train -o train -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-profile > $R/gpurun_out/rocprof_train.log 2>&1
cd $R
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -3; cut -c1-250 gpurun_out/bench_rccl1.json; tail -2 gpurun_out/bench_rccl1.err; cut -c1-250 gpurun_out/bench_train.json; echo; cut -c1-250 gpurun_out/bench_infer.json; echo; grep -E "BK32|\| 64x128  \|" gpurun_out/conv_bench.md
