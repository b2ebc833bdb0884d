#!/usr/bin/env bash
# round 3, visit 10: inference replicas (serving.ReplicaPool) test + bench legs
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_pipeline.py -m gpu -q --tb=short -p no:cacheprovider > $G/v10_pytest.log 2>&1; echo "pytest rc=$?" >> $G/v10_pytest.log
tail -4 $G/v10_pytest.log
for r in 1 2 3; do
  timeout 300 python bench.py --mode infer --replicas $r --steps 60 --warmup 10 --no-cpu-baseline --no-profile > $G/v10_infer_r$r.json 2> $G/v10_infer_r$r.err
  python -c "
import json;d=json.load(open('$G/v10_infer_r$r.json'));print('infer replicas=$r',round(d['value'],1),round(d['ms_per_step'],3), d['config'].get('single_replica'))" 2>/dev/null || { echo "r$r FAILED"; tail -5 $G/v10_infer_r$r.err; }
done
timeout 600 python bench.py --steps 50 --warmup 10 > $G/v10_full.json 2> $G/v10_full.err; python -c "
import json;d=json.load(open('$G/v10_full.json'));print('train',round(d['value'],1),'fv',round(d['forward_vote']['value'],1),d['forward_vote']['single_replica'],'cpu',d['cpu_baseline']['value'],d['cpu_baseline']['min'],d['cpu_baseline']['max'],'fvcpu',d['forward_vote']['cpu_baseline']['value']); print(d['roofline']['traffic'], d['roofline']['traffic_provenance'])" || tail -5 $G/v10_full.err
