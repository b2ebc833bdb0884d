#!/usr/bin/env bash
# round 3, visit 8: full GPU suite with the two-slot pipeline as the Engine default; A/B of the side wgrad stream / BN grids under it
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -f $G/test_branches.jsonl
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $G/v8_pytest.log 2>&1; echo "pytest rc=$?" >> $G/v8_pytest.log
grep -E "passed|failed|FAILED|rc=" $G/v8_pytest.log | tail -8
Q="--no-cpu-baseline --no-profile --no-forward-vote --steps 50 --warmup 10"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > $G/v8_$name.json 2> $G/v8_$name.err; python -c "
import json;d=json.load(open('$G/v8_$name.json'));print('$name',round(d['value'],1),round(d['ms_per_step'],3))" 2>/dev/null || { echo "$name FAILED"; tail -5 $G/v8_$name.err; }; }
run base DR_PIPELINE=2
run nows DR_PIPELINE=2 DR_WGRAD_STREAM=0
run nogroup DR_PIPELINE=2 DR_GROUP_WGRAD=0
run bn256 DR_PIPELINE=2 DR_BN_GRID=256
run bn1024 DR_PIPELINE=2 DR_BN_GRID=1024
run bn2048 DR_PIPELINE=2 DR_BN_GRID=2048
run elt1024 DR_PIPELINE=2 DR_ELT_GRID=1024
run base2 DR_PIPELINE=2
timeout 300 python bench.py --steps 100 --warmup 10 > $G/v8_full.json 2> $G/v8_full.err; cut -c1-600 $G/v8_full.json
