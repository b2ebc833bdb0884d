#!/usr/bin/env bash
# round 3, visit 36: the whole gpu suite at the round's last commit, smoke, and the headline bench lines three times each
# (run-to-run spread on one box)
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -3
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
T="--no-cpu-baseline --no-profile --no-forward-vote --steps 100 --warmup 10"
for i in 1 2 3; do
  timeout 300 python bench.py $T 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train', round(d['value'],1))"
  timeout 300 python bench.py $T --mode infer 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('infer', round(d['value'],1))"
done
