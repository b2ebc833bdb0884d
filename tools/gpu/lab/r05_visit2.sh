#!/usr/bin/env bash
# round 5, visit 2: conv_x3.h -- kernel tests, micro-benchmark against the fp32-MFMA kernels, effect on the training / inference step
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 300 python -m pytest tests/test_forward_parity.py -q -m gpu -k "x3" -p no:cacheprovider > gpurun_out/v2_x3_tests.log 2>&1; echo "rc=$?" >> gpurun_out/v2_x3_tests.log
timeout 600 python tools/x3_bench.py 200 > gpurun_out/v2_x3_bench_b200.md 2> gpurun_out/v2_x3_bench_b200.err
timeout 300 python tools/x3_bench.py 40 > gpurun_out/v2_x3_bench_b40.md 2> gpurun_out/v2_x3_bench_b40.err
for x in 0 1; do
  DR_CONV_X3=$x timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-forward-vote --detail gpurun_out/v2_detail_train_x3_$x.md > gpurun_out/v2_bench_train_x3_$x.json 2> gpurun_out/v2_bench_train_x3_$x.err
  DR_CONV_X3=$x timeout 400 python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline --detail gpurun_out/v2_detail_infer_x3_$x.md > gpurun_out/v2_bench_infer_x3_$x.json 2> gpurun_out/v2_bench_infer_x3_$x.err
done
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_bench_shapes.py -q -m gpu -p no:cacheprovider > gpurun_out/v2_parity_with_x3.log 2>&1; echo "rc=$?" >> gpurun_out/v2_parity_with_x3.log
tail -3 gpurun_out/v2_x3_tests.log; cat gpurun_out/v2_x3_bench_b200.md; tail -3 gpurun_out/v2_x3_bench_b200.err; cat gpurun_out/v2_x3_bench_b40.md
for x in 0 1; do cut -c1-200 gpurun_out/v2_bench_train_x3_$x.json; echo; tail -2 gpurun_out/v2_bench_train_x3_$x.err; cut -c1-200 gpurun_out/v2_bench_infer_x3_$x.json; echo; done
tail -5 gpurun_out/v2_parity_with_x3.log
