#!/usr/bin/env bash
# round 6, visit 10: the CLI training driver end to end (producer thread one window ahead, losses read one window late) next to bench.py
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_host_mirror.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile"
timeout 300 python bench.py $Q > gpurun_out/r06v10_bench.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r06v10_bench.json')); print('bench', round(d['value'],1))"
cd /tmp
for aug in True False; do
  timeout 600 python -m densereg_amd.model.hourglass_um_crop_tiny --dataset nyu --num_stack 2 --num_fea 128 --is_train True --is_aug $aug --max_steps 90 --synthetic_crops 2000 2>&1 | grep "^\[train\]\|Error\|error" | tail -3
done 2>&1 | tee $R/gpurun_out/r06v10_cli.log
