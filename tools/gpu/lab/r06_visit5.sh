#!/usr/bin/env bash
# round 6, visit 5: conv_x3h.h (3x3, haloed tile resident in LDS, weights by LDS-DMA) -- bit-identity on the GPU, time per launch, the training step
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_forward_parity.py -q -m gpu -k "conv_x3" -p no:cacheprovider -x > gpurun_out/r06v5_parity.log 2>&1; echo "rc=$?" >> gpurun_out/r06v5_parity.log
grep -v "start\]\|passed\]" gpurun_out/r06v5_parity.log | tail -8
timeout 600 python tools/x3h_bench.py 200 2>/dev/null | tee gpurun_out/r06v5_x3h_bench.md
Q="--steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > gpurun_out/r06v5_$name.json 2> gpurun_out/r06v5_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/r06v5_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)"; }
run halo_1 A=1
run nohalo_1 DR_X3_HALO=0
run halo_2 A=1
run nohalo_2 DR_X3_HALO=0
tail -5 gpurun_out/r06v5_halo_1.err
