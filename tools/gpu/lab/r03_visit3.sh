#!/usr/bin/env bash
# round 3, visit 3: cache policy of the output stores (dirty L2 lines are written back at every kernel boundary): plain vs
# nt vs write-through (sc1 / sc0 sc1) in the conv epilogue, the BatchReNorm apply passes and the weight-gradient slabs
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
SH="32:512:512:1:1 32:256:256:3:1 32:256:512:1:1 32:128:128:3:3 32:256:128:1:3 32:128:128:1:3 32:80:80:3:4 16:128:128:3"
rm -f $G/v3_probe.md
for v in base c1 c2 c3; do
  echo "## variant=$v" >> $G/v3_probe.md
  if [ $v = base ]; then timeout 200 python tools/conv_probe.py $SH >> $G/v3_probe.md 2>> $G/v3_probe.err
  else DR_LIB_VARIANT=$v timeout 200 python tools/conv_probe.py $SH >> $G/v3_probe.md 2>> $G/v3_probe.err; fi
done
Q="--no-cpu-baseline --no-profile --no-forward-vote --steps 40 --warmup 8"
for v in base c1 c2 c3 b1 b2 w2 all1 all2 base; do
  if [ $v = base ]; then timeout 200 python bench.py $Q > $G/v3_train_$v.json 2> $G/v3_train_$v.err
  else DR_LIB_VARIANT=$v timeout 200 python bench.py $Q > $G/v3_train_$v.json 2> $G/v3_train_$v.err; fi
  python -c "
import json;d=json.load(open('$G/v3_train_$v.json'));print('train $v',round(d['value'],1),round(d['ms_per_step'],3))" 2>/dev/null || { echo "$v FAILED"; tail -3 $G/v3_train_$v.err; }
done
for v in base c2 all2; do
  if [ $v = base ]; then timeout 200 python bench.py --mode infer $Q > $G/v3_infer_$v.json 2> $G/v3_infer_$v.err
  else DR_LIB_VARIANT=$v timeout 200 python bench.py --mode infer $Q > $G/v3_infer_$v.json 2> $G/v3_infer_$v.err; fi
  python -c "
import json;d=json.load(open('$G/v3_infer_$v.json'));print('infer $v',round(d['value'],1),round(d['ms_per_step'],3))" 2>/dev/null || { echo "$v FAILED"; tail -3 $G/v3_infer_$v.err; }
done
python - <<'PY'
import re
rows = {}; cur = None; order = []
for ln in open('gpurun_out/v3_probe.md'):
    m = re.match(r'## variant=(\w+)', ln)
    if m: cur = m.group(1); order.append(cur); continue
    f = [x.strip() for x in ln.strip().strip('|').split('|')]
    if len(f) == 8 and f[0].isdigit():
        rows.setdefault(tuple(f[:5]), {})[cur] = f[6]
print('shape | ' + ' | '.join(order))
for k, v in rows.items():
    print(':'.join(k) + ' | ' + ' | '.join(v.get(o, '-') for o in order))
PY
