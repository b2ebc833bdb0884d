#!/usr/bin/env bash
# round 4 visit 15: rows in flight per thread of the BatchReNorm passes (DR_BN_ROWS, build variants) on the bf16 path, where the raw loads are 8 bytes
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; G=gpurun_out
b() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > $G/r04_v15_$name.json 2> $G/r04_v15_$name.err; python -c "import json; d=json.load(open('$G/r04_v15_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))"; }
Q="--no-cpu-baseline --no-forward-vote --steps 40 --warmup 10 --no-profile --precision bf16"
b bf16_rows4 A=1
b bf16_rows8 DR_LIB_VARIANT=rows8
b bf16_rows2 DR_LIB_VARIANT=rows2
b bf16_rows4_2 A=1
b bf16_rows8_2 DR_LIB_VARIANT=rows8
Q="--no-cpu-baseline --no-forward-vote --steps 40 --warmup 10 --no-profile"
b f32_rows4 A=1
b f32_rows8 DR_LIB_VARIANT=rows8
Q="--num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu --no-cpu-baseline --steps 10 --warmup 3 --precision bf16 --no-forward-vote --no-profile"
b c5_rows4 A=1
b c5_rows8 DR_LIB_VARIANT=rows8
