#!/usr/bin/env bash
# GPU visit: BASELINE config 5 on one GPU -- NYU 4-stack fea=256, 256x256 crops, bf16 matrix cores (and fp32 beside it)
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
C5="--num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu --batch ${C5_BATCH:-8} --no-cpu-baseline"
timeout 120 python bench.py --mode train --precision bf16 $C5 --steps 5 --warmup 2 --detail gpurun_out/detail_c5_train_bf16.md > gpurun_out/bench_c5_train_bf16.json 2> gpurun_out/bench_c5_train_bf16.err; echo "rc=$?"; cut -c1-330 gpurun_out/bench_c5_train_bf16.json; echo
timeout 120 python bench.py --mode train --precision f32 $C5 --steps 5 --warmup 2 --no-profile > gpurun_out/bench_c5_train_f32.json 2> gpurun_out/bench_c5_train_f32.err; echo "rc=$?"; cut -c1-230 gpurun_out/bench_c5_train_f32.json; echo
timeout 120 python bench.py --mode infer --precision bf16 $C5 --steps 5 --warmup 2 --no-profile > gpurun_out/bench_c5_infer_bf16.json 2> gpurun_out/bench_c5_infer_bf16.err; echo "rc=$?"; cut -c1-230 gpurun_out/bench_c5_infer_bf16.json; echo
tail -2 gpurun_out/bench_c5_train_bf16.err
