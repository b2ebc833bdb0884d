#!/usr/bin/env bash
# round 3, visit 41: the loss kernel with one thread per (pixel, joint): parity (every training test), its time, the training step
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_train_parity.py tests/test_groups.py tests/test_gpu_configs.py tests/test_pipeline.py -m gpu -x -q 2>&1 | tail -2
T="--no-cpu-baseline --no-forward-vote --steps 100 --warmup 10"
for i in 1 2; do timeout 300 python bench.py $T 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']['all_kernels']; print('train', round(d['value'],1), 'loss ms/window', round(r['loss']['ms_per_step'],3), r['loss']['gbs'])"; done
timeout 300 python bench.py $T --groups 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']['all_kernels']; print('train g1', round(d['value'],1), 'loss ms/step', round(r['loss']['ms_per_step'],3))"
