#!/usr/bin/env bash
# round 6, visit 26: one 160-column x3 block for the 129..160-channel layers (conv_x3_kernel<128, 160>) against the fp32 16-column tiles (DR_X3_BN160=0)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_forward_parity.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -2
for m in 1 0; do echo "DR_X3_BN160=$m"; DR_X3_BN160=$m PROBE_B=200 timeout 300 python tools/conv_probe.py 32:128:131:1 32:256:156:1 32:512:156:1 2>/dev/null | tail -5; DR_X3_BN160=$m PROBE_B=40 timeout 300 python tools/conv_probe.py 64:142:142:3 64:129:129:3 64:284:142:1 2>/dev/null | tail -5; done | tee gpurun_out/r06v26_bn160.md
Q="--steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > gpurun_out/r06v26_$name.json 2> gpurun_out/r06v26_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/r06v26_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)"; }
for i in 1 2; do
run new_$i A=1
run old_$i DR_X3_BN160=0
done
Q="--steps 6 --warmup 3 --no-cpu-baseline --no-forward-vote --no-profile --num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu"
run c5_new A=1
run c5_old DR_X3_BN160=0
