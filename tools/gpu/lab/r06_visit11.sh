#!/usr/bin/env bash
# round 6, visit 11: the weight gradient of 515 -> 512 / 131 -> 128 as 128-channel x3 tiles + a tail kernel (DR_WG_TAIL) -- tests, step A/B, per-layer
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_train_parity.py tests/test_bench_shapes.py -q -m gpu -p no:cacheprovider -x -k "tail_split or wgrad or window or config" 2>&1 | tail -3
Q="--steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > gpurun_out/r06v11_$name.json 2> gpurun_out/r06v11_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/r06v11_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)"; }
run tail_1 A=1
run notail_1 DR_WG_TAIL=0
run tail_2 A=1
run notail_2 DR_WG_TAIL=0
run tail_3 A=1
run notail_3 DR_WG_TAIL=0
timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --detail gpurun_out/r06v11_detail.md > gpurun_out/r06v11_prof.json 2>/dev/null
grep "wgrad" gpurun_out/r06v11_detail.md | head -14
