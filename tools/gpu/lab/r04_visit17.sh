#!/usr/bin/env bash
# round 4 visit 17: the bf16-storage copies of the epilogue only in dedicated instantiations (XB bit 1): eval-mode bf16 forward + vote
# against record 5's build again, the bf16 training lines, the bf16 tests
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; G=gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "bf16 or config5" > $G/r04_v17_tests.log 2>&1; echo "rc=$?" >> $G/r04_v17_tests.log; tail -3 $G/r04_v17_tests.log
b() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > $G/r04_v17_$name.json 2> $G/r04_v17_$name.err; python -c "import json; d=json.load(open('$G/r04_v17_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))"; }
Q="--mode infer --precision bf16 --no-cpu-baseline --no-profile --steps 100 --warmup 10"
b infer_bf16_new A=1
b infer_bf16_rec5 DR_LIB_VARIANT=rec5
b infer_bf16_new2 A=1
b infer_bf16_rec5_2 DR_LIB_VARIANT=rec5
Q="--mode infer --precision bf16 --replicas 1 --merge 1 --no-cpu-baseline --no-profile --steps 100 --warmup 10"
b infer1_bf16_new A=1
b infer1_bf16_rec5 DR_LIB_VARIANT=rec5
b infer1_bf16_new2 A=1
b infer1_bf16_rec5_2 DR_LIB_VARIANT=rec5
Q="--no-cpu-baseline --no-forward-vote --steps 40 --warmup 10 --no-profile --precision bf16"
b train_bf16 A=1
b train_bf16_2 A=1
Q="--num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu --no-cpu-baseline --steps 10 --warmup 3 --precision bf16 --no-forward-vote --no-profile"
b c5 A=1
