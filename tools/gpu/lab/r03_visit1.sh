#!/usr/bin/env bash
# round 3, visit 1: full GPU suite (new: config-5 bf16 training tests, per-rank dropout, RAM-independent gradient bar) and the
# first experiment on the conv launch lock-step: static per-workgroup wave priorities (DR_CONV_PRIO / DR_WGRAD_PRIO)
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -f $G/test_branches.jsonl
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > $G/v1_pytest.log 2>&1; echo "pytest rc=$?" >> $G/v1_pytest.log
SH="32:512:512:1:1 32:256:256:3:1 32:256:512:1:1 32:512:256:1:1 32:128:128:3:3 32:256:128:1:3 32:128:128:1:3 32:80:80:3:4 32:160:80:1:4 32:131:65:1:4"
for pr in 0 1 2 3 4 5; do
  echo "## DR_CONV_PRIO=$pr" >> $G/v1_probe.md
  DR_CONV_PRIO=$pr timeout 200 python tools/conv_probe.py $SH >> $G/v1_probe.md 2>> $G/v1_probe.err
done
Q="--no-cpu-baseline --no-profile --no-forward-vote --steps 40 --warmup 8"
for pr in 0 1 2 3 4; do
  DR_CONV_PRIO=$pr timeout 200 python bench.py $Q > $G/v1_train_p$pr.json 2> $G/v1_train_p$pr.err
  DR_CONV_PRIO=$pr timeout 200 python bench.py --mode infer $Q > $G/v1_infer_p$pr.json 2> $G/v1_infer_p$pr.err
done
for wp in 1 2 3; do
  DR_WGRAD_PRIO=$wp timeout 200 python bench.py $Q > $G/v1_train_w$wp.json 2> $G/v1_train_w$wp.err
  DR_CONV_PRIO=1 DR_WGRAD_PRIO=$wp timeout 200 python bench.py $Q > $G/v1_train_p1w$wp.json 2> $G/v1_train_p1w$wp.err
done
grep -E "passed|failed|error" $G/v1_pytest.log | tail -5
grep -E "grad error|bf16 gradient" $G/v1_pytest.log | tail -12
cat $G/test_branches.jsonl 2>/dev/null | tail -8
cat $G/v1_probe.md
for f in train_p0 train_p1 train_p2 train_p3 train_p4 infer_p0 infer_p1 infer_p2 infer_p3 infer_p4 train_w1 train_w2 train_w3 train_p1w1 train_p1w2 train_p1w3; do python -c "
import json;d=json.load(open('$G/v1_$f.json'));print('$f',round(d['value'],1),round(d['ms_per_step'],3))" 2>/dev/null || { echo "$f FAILED"; tail -3 $G/v1_$f.err; }; done
