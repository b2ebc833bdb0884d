#!/usr/bin/env bash
# GPU visit: gpu tests, bench (train/infer) with per-layer detail, rocprof kernel stats, PMC passes (FETCH_SIZE / WRITE_SIZE)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
DR_CONV_GLDS=1 timeout 600 python -m pytest tests/test_forward_parity.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu_glds.log 2>&1; echo "pytest(glds) rc=$?" >> gpurun_out/pytest_gpu_glds.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 --detail gpurun_out/detail_train.md > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err; echo "bench rc=$?" >> gpurun_out/bench_train.err
timeout 600 python bench.py --mode infer --steps 20 --warmup 5 --detail gpurun_out/detail_infer.md > gpurun_out/bench_infer.json 2> gpurun_out/bench_infer.err
for prec in bf16; do
  timeout 300 python bench.py --mode infer --precision $prec --steps 40 --warmup 10 --no-cpu-baseline --detail gpurun_out/detail_infer_$prec.md > gpurun_out/bench_infer_$prec.json 2> gpurun_out/bench_infer_$prec.err
  timeout 300 python bench.py --mode train --precision $prec --steps 20 --warmup 5 --no-cpu-baseline --detail gpurun_out/detail_train_$prec.md > gpurun_out/bench_train_$prec.json 2> gpurun_out/bench_train_$prec.err
done
# the launch path the driver uses for N > 1 (one rank here) and the all-reduce leg on a single GPU
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-profile > gpurun_out/bench_torchrun.json 2> gpurun_out/bench_torchrun.err; echo "torchrun rc=$?" >> gpurun_out/bench_torchrun.err
DR_FORCE_ALLREDUCE=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile > gpurun_out/bench_allreduce.json 2> gpurun_out/bench_allreduce.err; echo "allreduce rc=$?" >> gpurun_out/bench_allreduce.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train -o train -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-profile > $R/gpurun_out/rocprof_train.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_infer -o infer -- python $R/bench.py --mode infer --steps 10 --warmup 5 --no-cpu-baseline --no-profile > $R/gpurun_out/rocprof_infer.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $R/gpurun_out/pmc_fetch.log 2>&1; echo "rc=$?" >> $R/gpurun_out/pmc_fetch.log
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $R/gpurun_out/pmc_write.log 2>&1; echo "rc=$?" >> $R/gpurun_out/pmc_write.log
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_infer -o fetch -- python $R/bench.py --mode infer --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $R/gpurun_out/pmc_fetch_infer.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_infer -o write -- python $R/bench.py --mode infer --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $R/gpurun_out/pmc_write_infer.log 2>&1
cd $R
grep -E "passed|failed" gpurun_out/pytest_gpu.log gpurun_out/pytest_gpu_glds.log | tail -4; cut -c1-120 gpurun_out/bench_torchrun.json; tail -1 gpurun_out/bench_torchrun.err; cut -c1-120 gpurun_out/bench_allreduce.json; tail -1 gpurun_out/bench_allreduce.err; tail -2 gpurun_out/smoke.log; cut -c1-330 gpurun_out/bench_train.json; echo; cut -c1-330 gpurun_out/bench_infer.json; echo; cut -c1-200 gpurun_out/bench_infer_bf16.json; echo; cut -c1-200 gpurun_out/bench_train_bf16.json; echo; ls -la gpurun_out/pmc_fetch gpurun_out/pmc_write; tail -3 gpurun_out/pmc_fetch.log
