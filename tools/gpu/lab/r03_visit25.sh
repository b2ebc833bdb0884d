#!/usr/bin/env bash
# round 3, visit 25: the whole gpu suite with micro-batch groups in; forward(eval)+vote at larger batches per launch (is batching
# submitted batches worth more than replicas?); bf16 training with groups, config 5 with a forced window
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -8
Q="--no-cpu-baseline --no-profile --steps 100 --warmup 10"
run() { name=$1; shift; env "$@" > $G/v25_$name.json 2> $G/v25_$name.err; python -c "
import json;d=json.load(open('$G/v25_$name.json'));print('$name',round(d['value'],1),round(d['ms_per_step'],3),(d['config'].get('single_replica') or {}).get('value'))" 2>/dev/null || { echo "$name FAILED"; tail -5 $G/v25_$name.err; }; }
run infer40x3 timeout 300 python bench.py $Q --mode infer
run infer80x1 timeout 300 python bench.py $Q --mode infer --batch 80 --replicas 1
run infer80x2 timeout 300 python bench.py $Q --mode infer --batch 80 --replicas 2
run infer120x1 timeout 300 python bench.py $Q --mode infer --batch 120 --replicas 1
run infer120x2 timeout 300 python bench.py $Q --mode infer --batch 120 --replicas 2
run infer200x1 timeout 300 python bench.py $Q --mode infer --batch 200 --replicas 1
run infer200x2 timeout 300 python bench.py $Q --mode infer --batch 200 --replicas 2
run c5_bf16_g5 timeout 400 python bench.py --num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu --no-cpu-baseline --no-profile --no-forward-vote --steps 10 --warmup 5 --precision bf16 --groups 5
run c5_bf16_g1 timeout 400 python bench.py --num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu --no-cpu-baseline --no-profile --no-forward-vote --steps 10 --warmup 5 --precision bf16
