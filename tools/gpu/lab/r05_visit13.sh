#!/usr/bin/env bash
# round 5, last visit: what the driver runs at round end, on the final tree -- pytest -m gpu, smoke(), bench.py with its defaults
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -f gpurun_out/pytest_live.log
( time timeout 1100 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/v13_suite.log 2>&1; echo "rc=$?" >> gpurun_out/v13_suite.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/v13_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/v13_smoke.log
( time timeout 600 python bench.py ) > gpurun_out/v13_bench_default.json 2> gpurun_out/v13_bench_default.err; echo "bench rc=$?" >> gpurun_out/v13_bench_default.err
grep -v "start\]\|passed\]" gpurun_out/v13_suite.log | tail -6; tail -2 gpurun_out/v13_smoke.log; tail -5 gpurun_out/v13_bench_default.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/v13_bench_default.json').read().strip().splitlines()[0])
r=d['roofline']
print(d['metric'], round(d['value'],1), d['unit'], d['n_gpus'], d['steps'], d['warmup'], round(d['ms_per_step'],3), d['dtype'], d['scaling'], d['vs_baseline'])
print('roofline', r['kernel'], round(r['achieved'],1), r['peak'], round(r['frac'],3), r['traffic'], r['peak_basis'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['kind'])
print('forward_vote', round(d['forward_vote']['value'],1))
PY
