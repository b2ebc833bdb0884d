#!/usr/bin/env bash
# round 6, visit 13: 256-column blocks in conv_x3_kernel (DR_X3_BIG = 1: eight waves of 64x64, 2: sixteen waves of 64x32) on the wide 1x1 layers
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for m in 3 4; do echo "== parity DR_X3_BIG=$m"; DR_X3_BIG=$m timeout 600 python -m pytest tests/test_forward_parity.py -q -m gpu -k "conv_x3" -p no:cacheprovider -x 2>&1 | tail -2; done
for m in 0 3 4 0 3 4; do DR_X3_BIG=$m timeout 300 python tools/x3_bn256_bench.py 200 2>/dev/null; done | tee gpurun_out/r06v13_bm256.md
Q="--steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > gpurun_out/r06v13_$name.json 2> gpurun_out/r06v13_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/r06v13_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)"; }
run base_1 A=1
run m8_1 DR_X3_BIG=3
run m16_1 DR_X3_BIG=4
run base_2 A=1
run m8_2 DR_X3_BIG=3
run m16_2 DR_X3_BIG=4
