#!/usr/bin/env bash
# round 5, visit 10: the driver's pytest command once more on a fresh box; forward+vote by replicas x merged batches
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -f gpurun_out/pytest_live.log
( time timeout 1100 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/v10_suite.log 2>&1; echo "rc=$?" >> gpurun_out/v10_suite.log
for r in 1 2 3; do for m in 1 5 8 10; do
  timeout 200 python bench.py --mode infer --replicas $r --merge $m --steps 40 --warmup 10 --no-cpu-baseline --no-profile > gpurun_out/v10_infer_r${r}_m${m}.json 2> gpurun_out/v10_infer_r${r}_m${m}.err
  python -c "
import json
try:
    d=json.load(open('gpurun_out/v10_infer_r${r}_m${m}.json')); print('replicas $r merge $m', round(d['value'],1))
except Exception as e: print('replicas $r merge $m failed', e)"
done; done
grep -v "start\]\|passed\]" gpurun_out/v10_suite.log | tail -8
