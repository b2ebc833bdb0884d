#!/usr/bin/env bash
# round 4 visit 11: raw outputs of BatchReNorm convs stored as bf16 on the bf16 path (DR_BF16_RAW): bf16 tests, A/B on both bf16 lines,
# the fp32 headline beside it (its kernels are the same code)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; G=gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "bf16 or bn_layer or config5" > $G/r04_v11_tests.log 2>&1; echo "rc=$?" >> $G/r04_v11_tests.log; tail -4 $G/r04_v11_tests.log
Q="--no-cpu-baseline --no-forward-vote --steps 40 --warmup 10 --no-profile"
b() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > $G/r04_v11_$name.json 2> $G/r04_v11_$name.err; python -c "import json; d=json.load(open('$G/r04_v11_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))"; }
Q="$Q --precision bf16"
b bf16_raw16 A=1
b bf16_raw32 DR_BF16_RAW=0
b bf16_raw16_2 A=1
b bf16_raw32_2 DR_BF16_RAW=0
Q="--num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu --no-cpu-baseline --steps 10 --warmup 3 --precision bf16 --no-forward-vote --no-profile"
b c5_raw16 A=1
b c5_raw32 DR_BF16_RAW=0
Q="--no-cpu-baseline --no-forward-vote --steps 40 --warmup 10 --no-profile"
b f32 A=1
