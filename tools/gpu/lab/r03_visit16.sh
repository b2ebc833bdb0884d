#!/usr/bin/env bash
# round 3, visit 16: slot / replica stream priorities
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for p in 0 1 2 1; do
  DR_PIPE_PRIO=$p timeout 400 python bench.py --no-cpu-baseline --no-profile --steps 50 --warmup 10 > $G/v16_p$p.json 2> $G/v16_p$p.err
  python -c "
import json;d=json.load(open('$G/v16_p$p.json'));print('slot prio mode=$p train',round(d['value'],1),'fwd+vote pool',round(d['forward_vote']['value'],1),'single',round(d['forward_vote']['single_replica']['value'],1))" 2>/dev/null || { echo "p$p FAILED"; tail -5 $G/v16_p$p.err; }
done
for r in 2 3; do timeout 400 python bench.py --mode infer --replicas $r --no-cpu-baseline --no-profile --steps 60 --warmup 10 > $G/v16_i$r.json 2> $G/v16_i$r.err; python -c "
import json;d=json.load(open('$G/v16_i$r.json'));print('infer x$r',round(d['value'],1))"; done
DR_PIPE_PRIO=1 timeout 300 python bench.py --no-cpu-baseline --no-profile --no-forward-vote --steps 50 --warmup 10 --precision bf16 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('bf16 prio1',round(d['value'],1))"
python tests/stress_pipeline.py 4 2>&1 | grep depth | cut -c1-120
