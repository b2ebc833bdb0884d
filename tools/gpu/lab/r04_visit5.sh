#!/usr/bin/env bash
# round 4 visit 5: fused hourglass bottoms after the cross-convolution prefetch and the per-width column tiles
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; G=gpurun_out
timeout 600 python -m pytest tests/test_fused_tail.py tests/test_forward_parity.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider > $G/r04_v5_tests.log 2>&1; echo "rc=$?" >> $G/r04_v5_tests.log
tail -4 $G/r04_v5_tests.log
LAT_BATCHES=1,8,40 timeout 600 python tools/latency_bench.py > $G/r04_v5_latency.md 2>&1; cat $G/r04_v5_latency.md
timeout 300 python bench.py --mode infer --replicas 1 --merge 1 --no-cpu-baseline --steps 40 --warmup 10 --detail $G/r04_v5_detail_infer.md > $G/r04_v5_infer_prof.json 2> $G/r04_v5_infer_prof.err; grep hourglass $G/r04_v5_detail_infer.md
timeout 200 python bench.py --no-cpu-baseline --steps 100 --warmup 10 --no-profile --mode infer > $G/r04_v5_pool.json 2> $G/r04_v5_pool.err; python -c "import json; d=json.load(open('$G/r04_v5_pool.json')); print('pool', round(d['value'],1), d['config']['single_replica'])"
