#!/usr/bin/env bash
# round 6, visit 16: conv_x3_kernel's pixel path -- DR_X3_PF = 1 (two K-tiles of pixels in flight), 2 (scheduling barrier between a K-tile's
# MFMAs and the split of the next tile: hipcc hoists the split, and the wait for its global load, in among the first MFMAs of every other tile)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for m in 1 2; do echo "== parity DR_X3_PF=$m"; DR_X3_PF=$m timeout 600 python -m pytest tests/test_forward_parity.py -q -m gpu -k "conv_x3" -p no:cacheprovider -x 2>&1 | tail -2; done
for m in 0 1 2 0 1 2; do echo "DR_X3_PF=$m"; DR_X3_PF=$m timeout 300 python tools/x3_bn256_bench.py 200 2>/dev/null; done | tee gpurun_out/r06v16_pf.md
Q="--steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > gpurun_out/r06v16_$name.json 2> gpurun_out/r06v16_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/r06v16_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)"; }
run base_1 A=1
run pf1_1 DR_X3_PF=1
run pf2_1 DR_X3_PF=2
run base_2 A=1
run pf1_2 DR_X3_PF=1
run pf2_2 DR_X3_PF=2
