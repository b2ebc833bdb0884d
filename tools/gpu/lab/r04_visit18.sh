#!/usr/bin/env bash
# round 4 visit 18: rows in flight per thread of the BatchReNorm passes once more: 12 / 16 for the bf16-raw variants, 3 / 6 for fp32
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; G=gpurun_out
b() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > $G/r04_v18_$name.json 2> $G/r04_v18_$name.err; python -c "import json; d=json.load(open('$G/r04_v18_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))"; }
Q="--no-cpu-baseline --no-forward-vote --steps 40 --warmup 10 --no-profile --precision bf16"
b bf16_r8 A=1
b bf16_r12 DR_LIB_VARIANT=r12
b bf16_r16 DR_LIB_VARIANT=r16
b bf16_r8_2 A=1
b bf16_r12_2 DR_LIB_VARIANT=r12
b bf16_r16_2 DR_LIB_VARIANT=r16
Q="--no-cpu-baseline --no-forward-vote --steps 40 --warmup 10 --no-profile"
b f32_r4 A=1
b f32_r6 DR_LIB_VARIANT=f6
b f32_r3 DR_LIB_VARIANT=f3
b f32_r4_2 A=1
b f32_r6_2 DR_LIB_VARIANT=f6
b f32_r3_2 DR_LIB_VARIANT=f3
Q="--num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu --no-cpu-baseline --steps 10 --warmup 3 --precision bf16 --no-forward-vote --no-profile"
b c5_r8 A=1
b c5_r12 DR_LIB_VARIANT=r12
b c5_r16 DR_LIB_VARIANT=r16
