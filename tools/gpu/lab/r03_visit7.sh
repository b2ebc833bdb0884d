#!/usr/bin/env bash
# round 3, visit 7: two micro-steps in flight (dr_set_pipeline): GPU tests + bench A/B
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_pipeline.py tests/test_host_mirror.py -m gpu -q --tb=short -p no:cacheprovider -x > $G/v7_pytest.log 2>&1; echo "pytest rc=$?" >> $G/v7_pytest.log
tail -5 $G/v7_pytest.log
Q="--no-cpu-baseline --no-profile --no-forward-vote --steps 50 --warmup 10"
for d in 1 2 2 1; do
  DR_PIPELINE=$d timeout 300 python bench.py $Q > $G/v7_train_d$d.json 2> $G/v7_train_d$d.err
  python -c "
import json;d=json.load(open('$G/v7_train_d$d.json'));print('train depth=$d',round(d['value'],1),round(d['ms_per_step'],3))" 2>/dev/null || { echo "d$d FAILED"; tail -5 $G/v7_train_d$d.err; }
done
for sb in 1 2 10; do
  DR_PIPELINE=2 timeout 300 python bench.py $Q --sub_batch $sb > $G/v7_train_sb$sb.json 2> $G/v7_train_sb$sb.err
  python -c "
import json;d=json.load(open('$G/v7_train_sb$sb.json'));print('train depth=2 sub_batch=$sb',round(d['value'],1),round(d['ms_per_step'],3))" 2>/dev/null || { echo "sb$sb FAILED"; tail -5 $G/v7_train_sb$sb.err; }
done
DR_PIPELINE=2 timeout 300 python bench.py $Q --dataset msra > $G/v7_msra.json 2> $G/v7_msra.err; python -c "
import json;d=json.load(open('$G/v7_msra.json'));print('msra depth=2',round(d['value'],1))"
DR_PIPELINE=2 timeout 300 python bench.py $Q --precision bf16 > $G/v7_bf16.json 2> $G/v7_bf16.err; python -c "
import json;d=json.load(open('$G/v7_bf16.json'));print('bf16 depth=2',round(d['value'],1))"
