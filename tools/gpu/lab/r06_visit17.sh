#!/usr/bin/env bash
# round 6, visit 17: co-resident workgroups out of phase (DR_X3_STAGGER = k: the second workgroup of every CU starts K-tiles * k / 16 x s_sleep(127) late, once)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for m in 0 2 3 5 8 0; do echo "DR_X3_STAGGER=$m"; DR_X3_STAGGER=$m timeout 300 python tools/x3_bn256_bench.py 200 2>/dev/null; done | tee gpurun_out/r06v17_stagger.md
Q="--steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > gpurun_out/r06v17_$name.json 2> gpurun_out/r06v17_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/r06v17_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)"; }
run base_1 A=1
run st3_1 DR_X3_STAGGER=3
run st5_1 DR_X3_STAGGER=5
run base_2 A=1
run st3_2 DR_X3_STAGGER=3
run st5_2 DR_X3_STAGGER=5
