#!/usr/bin/env bash
# round 4 visit 9: BatchReNorm-backward sums of a LEADING-slice producer (comb|uvd) from its reader's narrowed dgrad: A/B + tests
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; G=gpurun_out
timeout 600 python -m pytest tests/test_train_parity.py tests/test_groups.py tests/test_bn_layer.py tests/test_bench_shapes.py -m gpu -q --tb=short -p no:cacheprovider -k "not replica" > $G/r04_v9_tests.log 2>&1; echo "rc=$?" >> $G/r04_v9_tests.log; tail -3 $G/r04_v9_tests.log
Q="--no-cpu-baseline --no-forward-vote --steps 40 --warmup 10 --no-profile"
b() { name=$1; shift; env "$@" timeout 200 python bench.py $Q > $G/r04_v9_$name.json 2> $G/r04_v9_$name.err; python -c "import json; d=json.load(open('$G/r04_v9_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))"; }
b slice_on A=1
b slice_off DR_FUSE_SLICE=0
b slice_on2 A=1
b slice_off2 DR_FUSE_SLICE=0
