#!/usr/bin/env bash
# round 4 visit 16: eval-mode forward + vote on the bf16 matrix cores, this build against record 5's (variant "rec5"): did the
# three-copy epilogue of the bf16 conv kernels cost the inference path anything?
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; G=gpurun_out
b() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > $G/r04_v16_$name.json 2> $G/r04_v16_$name.err; python -c "import json; d=json.load(open('$G/r04_v16_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))"; }
Q="--mode infer --precision bf16 --no-cpu-baseline --no-profile --steps 100 --warmup 10"
b infer_bf16_new A=1
b infer_bf16_rec5 DR_LIB_VARIANT=rec5
b infer_bf16_new2 A=1
b infer_bf16_rec5_2 DR_LIB_VARIANT=rec5
Q="--mode infer --precision bf16 --replicas 1 --merge 1 --no-cpu-baseline --no-profile --steps 100 --warmup 10"
b infer1_bf16_new A=1
b infer1_bf16_rec5 DR_LIB_VARIANT=rec5
