#!/usr/bin/env bash
# round 6, visit 28: the bf16 path's BatchReNorm passes with non-temporal 8-byte loads / stores too (DR_BN_NT=15, build variant nt15)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { name=$1; shift; env "$@" timeout 400 python bench.py $Q > gpurun_out/r06v28_$name.json 2> gpurun_out/r06v28_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/r06v28_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)"; }
Q="--steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile --num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu --precision bf16"
for i in 1 2; do run c5_base_$i A=1; run c5_nt15_$i DR_LIB_VARIANT=nt15; done
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile --precision bf16"
for i in 1 2; do run s2_base_$i A=1; run s2_nt15_$i DR_LIB_VARIANT=nt15; done
Q="--steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile"
for i in 1 2; do run f32_base_$i A=1; run f32_r05_$i DR_X3_HALO=0 DR_X3_BD=0 DR_WG_TAIL=0 DR_X3_BN160=0; done
