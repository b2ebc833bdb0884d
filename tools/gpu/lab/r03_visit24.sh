#!/usr/bin/env bash
# round 3, visit 24: micro-batch groups (dr_set_groups): parity on the GPU, then the training step with the accumulation window as
# one pass of launches (G = sub_batch = 5, 200 crops per pass) against one micro-step per pass, at pipeline depth 1 and 2
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_groups.py -m gpu -x -q 2>&1 | tail -15
T="--no-cpu-baseline --no-profile --no-forward-vote --steps 100 --warmup 10"
run() { name=$1; shift; env "$@" > $G/v24_$name.json 2> $G/v24_$name.err; python -c "
import json;d=json.load(open('$G/v24_$name.json'));print('$name',round(d['value'],1),round(d['ms_per_step'],3),d['config'].get('micro_steps_per_pass'),d['config'].get('micro_steps_in_flight'))" 2>/dev/null || { echo "$name FAILED"; tail -5 $G/v24_$name.err; }; }
run g1 timeout 300 python bench.py $T --groups 1
run g5 timeout 300 python bench.py $T
run g5_depth2 DR_PIPELINE=2 timeout 300 python bench.py $T
run g5_nows DR_WGRAD_STREAM=0 timeout 300 python bench.py $T
run g5_bf16 timeout 300 python bench.py $T --precision bf16
run g1_bf16 timeout 300 python bench.py $T --precision bf16 --groups 1
run g5_msra timeout 300 python bench.py $T --dataset msra
timeout 300 python bench.py --no-cpu-baseline --no-forward-vote --steps 20 --warmup 5 --detail $G/v24_detail_g5.md > $G/v24_prof.json 2> $G/v24_prof.err; python -c "
import json;d=json.load(open('$G/v24_prof.json'));r=d['roofline'];print('roof',r['kernel'],round(r['frac'],3),round(r['avg_launch_us'],1));print({k:round(v['ms_per_step'],3) for k,v in r['all_kernels'].items()})"
head -40 $G/v24_detail_g5.md
