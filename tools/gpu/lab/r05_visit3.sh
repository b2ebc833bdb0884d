#!/usr/bin/env bash
# round 5, visit 3: x3 accuracy on hardware + 64-column tile, training step with / without x3 (both accumulator variants), first training run on synthetic hands
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 300 python -m pytest tests/test_forward_parity.py -q -m gpu -k "x3" -p no:cacheprovider > gpurun_out/v3_x3_tests.log 2>&1; echo "rc=$?" >> gpurun_out/v3_x3_tests.log
timeout 600 python tools/x3_bench.py 200 > gpurun_out/v3_x3_bench_b200.md 2> gpurun_out/v3_x3_bench_b200.err
for cfg in "0 0" "1 0" "1 1"; do
  set -- $cfg
  DR_CONV_X3=$1 DR_X3_VARIANT=$2 timeout 400 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --detail gpurun_out/v3_detail_train_x3_$1$2.md > gpurun_out/v3_bench_train_x3_$1$2.json 2> gpurun_out/v3_bench_train_x3_$1$2.err
done
DR_CONV_X3=1 DR_X3_BN64=1 timeout 400 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --detail gpurun_out/v3_detail_train_x3_bn64.md > gpurun_out/v3_bench_train_x3_bn64.json 2> gpurun_out/v3_bench_train_x3_bn64.err
timeout 600 python examples/train_synthetic.py --steps 300 > gpurun_out/v3_train_synth.log 2>&1; echo "rc=$?" >> gpurun_out/v3_train_synth.log
tail -3 gpurun_out/v3_x3_tests.log; cat gpurun_out/v3_x3_bench_b200.md; tail -3 gpurun_out/v3_x3_bench_b200.err
for x in 00 10 11 bn64; do cut -c1-160 gpurun_out/v3_bench_train_x3_$x.json; echo; tail -1 gpurun_out/v3_bench_train_x3_$x.err; done
tail -25 gpurun_out/v3_train_synth.log
