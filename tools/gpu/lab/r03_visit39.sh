#!/usr/bin/env bash
# round 3, visit 39: weight-gradient workgroup mapping -- within an XCD the (tile, tap) index fastest, the slab slowest (all
# workgroups of a slab start together on one L2) against the slab-fastest mapping; kernel sweep, parity, training step
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_train_parity.py tests/test_groups.py -m gpu -x -q -k "wgrad or groups or single_stack or config3" 2>&1 | tail -2
export PROBE_SHAPES="32:256:256:3,32:128:128:3,32:512:512:1,32:256:512:1,32:512:256:1,32:128:256:1,32:64:64:3" PROBE_NS="0,16,24,32,40,48,56,64,80,96,128" PROBE_T=128
PROBE_B=200 timeout 600 python tools/wgrad_bench.py > $G/v39_new_b200.md 2>&1
DR_WGRAD_SLAB_MAJOR=0 PROBE_B=200 timeout 600 python tools/wgrad_bench.py > $G/v39_old_b200.md 2>&1
PROBE_NS=0 PROBE_B=40 timeout 600 python tools/wgrad_bench.py > $G/v39_new_b40.md 2>&1
DR_WGRAD_SLAB_MAJOR=0 PROBE_NS=0 PROBE_B=40 timeout 600 python tools/wgrad_bench.py > $G/v39_old_b40.md 2>&1
python - <<PY
import collections
def load(f):
    d=collections.OrderedDict()
    for l in open('$G/'+f):
        c=[x.strip() for x in l.strip().strip('|').split('|')]
        if len(c)<8 or not c[0].isdigit(): continue
        d[(c[1],c[2],c[3],c[5])]=float(c[6])
    return d
for a,b in (('v39_old_b200.md','v39_new_b200.md'),('v39_old_b40.md','v39_new_b40.md')):
    o,n=load(a),load(b)
    print(a,'->',b)
    for k in o:
        if k in n: print(k, o[k], n[k], '%.1f%%'%(100*(n[k]/o[k]-1)))
PY
T="--no-cpu-baseline --no-profile --no-forward-vote --steps 100 --warmup 10"
for v in 0 1 0 1; do DR_WGRAD_SLAB_MAJOR=$v timeout 300 python bench.py $T 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train slab_major=$v', round(d['value'],1))"; done
