#!/usr/bin/env bash
# round 3, visit 15: hardware queues vs streams (GPU_MAX_HW_QUEUES), the forward+vote pool inside the train bench
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for q in default 2 4 8 16; do
  if [ $q = default ]; then E=""; else E="GPU_MAX_HW_QUEUES=$q"; fi
  env $E timeout 400 python bench.py --no-cpu-baseline --no-profile --steps 50 --warmup 10 > $G/v15_q$q.json 2> $G/v15_q$q.err
  python -c "
import json;d=json.load(open('$G/v15_q$q.json'));print('queues=$q train',round(d['value'],1),'fwd+vote pool',round(d['forward_vote']['value'],1),'single',round(d['forward_vote']['single_replica']['value'],1))" 2>/dev/null || { echo "q$q FAILED"; tail -5 $G/v15_q$q.err; }
done
for q in default 8; do
  if [ $q = default ]; then E=""; else E="GPU_MAX_HW_QUEUES=$q"; fi
  env $E timeout 400 python bench.py --mode infer --replicas 3 --no-cpu-baseline --no-profile --steps 60 --warmup 10 > $G/v15_iq$q.json 2> $G/v15_iq$q.err
  python -c "
import json;d=json.load(open('$G/v15_iq$q.json'));print('queues=$q infer x3',round(d['value'],1))"
done
