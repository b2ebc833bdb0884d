#!/usr/bin/env bash
# round 3, visit 31: the two headline bench lines again, now that profiles/pmc_traffic.json carries THIS build's PMC passes
# (roofline.traffic is quoted only for a matching kernel-source hash)
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 400 python bench.py --detail $G/r03_detail_train.md > $G/r03_bench_train.json 2> $G/r03_bench_train.err; echo "bench rc=$?"
timeout 300 python bench.py --mode infer --detail $G/r03_detail_infer.md > $G/r03_bench_infer.json 2> $G/r03_bench_infer.err; echo "bench rc=$?"
python - <<PY
import json
for f in ('train','infer'):
    d=json.load(open('$G/r03_bench_%s.json'%f)); r=d['roofline']
    print(f, round(d['value'],1), r['kernel'], round(r['frac'],3), r['traffic'], r['avg_launch_us'], r['algorithmic_gflop_per_launch'])
PY
