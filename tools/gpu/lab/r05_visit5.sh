#!/usr/bin/env bash
# round 5, visit 5: full GPU suite (driver's command), weight-gradient x3 micro-benchmark, training step with / without it
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -f gpurun_out/pytest_live.log
( time timeout 1100 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/v5_suite.log 2>&1; echo "rc=$?" >> gpurun_out/v5_suite.log
cp gpurun_out/pytest_live.log gpurun_out/v5_pytest_live.log
timeout 600 python tools/wgrad_x3_bench.py 200 > gpurun_out/v5_wgrad_x3_bench.md 2> gpurun_out/v5_wgrad_x3_bench.err
for w in 0 1; do
  DR_WGRAD_X3=$w timeout 400 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --detail gpurun_out/v5_detail_train_wx3_$w.md > gpurun_out/v5_bench_train_wx3_$w.json 2> gpurun_out/v5_bench_train_wx3_$w.err
done
DR_WGRAD_X3=1 DR_WGRAD_STREAM=0 timeout 400 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile > gpurun_out/v5_bench_train_wx3_inline.json 2> gpurun_out/v5_bench_train_wx3_inline.err
grep -v "start\]\|passed\]" gpurun_out/v5_suite.log | tail -12
cat gpurun_out/v5_wgrad_x3_bench.md; tail -3 gpurun_out/v5_wgrad_x3_bench.err
for w in 0 1 inline; do cut -c1-160 gpurun_out/v5_bench_train_wx3_$w.json; echo; tail -1 gpurun_out/v5_bench_train_wx3_$w.err; done
