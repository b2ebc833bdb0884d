#!/usr/bin/env bash
# round 3, visit 9: full GPU suite again (wgrad x_bf16 line restored), pipelined bf16 / config-5 figures
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -f $G/test_branches.jsonl
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $G/v9_pytest.log 2>&1; echo "pytest rc=$?" >> $G/v9_pytest.log
grep -E "passed|failed|FAILED|rc=" $G/v9_pytest.log | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
Q="--no-cpu-baseline --no-profile --no-forward-vote --steps 50 --warmup 10"
run() { name=$1; shift; env "$@" timeout 400 python bench.py $Q $EXTRA > $G/v9_$name.json 2> $G/v9_$name.err; python -c "
import json;d=json.load(open('$G/v9_$name.json'));print('$name',round(d['value'],1),round(d['ms_per_step'],3))" 2>/dev/null || { echo "$name FAILED"; tail -5 $G/v9_$name.err; }; }
EXTRA="--precision bf16"; run bf16_d1 DR_PIPELINE=1; run bf16_d2 DR_PIPELINE=2
EXTRA="--dataset msra"; run msra_d1 DR_PIPELINE=1; run msra_d2 DR_PIPELINE=2
Q="--no-cpu-baseline --no-profile --no-forward-vote --steps 12 --warmup 4"
EXTRA="--num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu --precision bf16"; run c5_bf16_d1 DR_PIPELINE=1; run c5_bf16_d2 DR_PIPELINE=2
EXTRA="--num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu"; run c5_f32_d1 DR_PIPELINE=1; run c5_f32_d2 DR_PIPELINE=2
rocm-smi --showmeminfo vram 2>/dev/null | tail -4
