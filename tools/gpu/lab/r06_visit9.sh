#!/usr/bin/env bash
# round 6, visit 9: conv_x3_kernel's weight tiles by a hidden LDS-DMA (BD) against the register-staged ones: 1x1 shapes at 200 crops, the step
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_forward_parity.py -q -m gpu -k "conv_x3" -p no:cacheprovider -x 2>&1 | tail -2
for bd in 1 0 1 0; do echo "== DR_X3_BD=$bd"; DR_X3_BD=$bd timeout 300 python tools/p3_bench.py 200 2>/dev/null | cut -c1-70 | sed -n 3,12p; done | tee gpurun_out/r06v9_bd.md
Q="--steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > gpurun_out/r06v9_$name.json 2> gpurun_out/r06v9_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/r06v9_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)"; }
run bd_1 A=1
run nobd_1 DR_X3_BD=0
run bd_2 A=1
run nobd_2 DR_X3_BD=0
