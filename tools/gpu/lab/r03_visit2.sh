#!/usr/bin/env bash
# round 3, visit 2: does de-phasing the co-resident workgroups of a conv launch pay?  DR_CONV_PRIO modes 6 / 7 (priority by
# hardware wave slot), 8 (fixed staggered start of the first round), 9 (stagger proportional to the layer's K-tiles)
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
SH="32:512:512:1:1 32:256:256:3:1 32:256:512:1:1 32:512:256:1:1 32:128:128:3:3 32:256:128:1:3 32:128:128:1:3 32:80:80:3:4 32:160:80:1:4"
rm -f $G/v2_probe.md
for pr in 0 6 7 24 40 72 136 264 520 41 73 137; do
  echo "## DR_CONV_PRIO=$pr" >> $G/v2_probe.md
  DR_CONV_PRIO=$pr timeout 200 python tools/conv_probe.py $SH >> $G/v2_probe.md 2>> $G/v2_probe.err
done
Q="--no-cpu-baseline --no-profile --no-forward-vote --steps 40 --warmup 8"
for pr in 0 6 40 72 136 73; do
  DR_CONV_PRIO=$pr timeout 200 python bench.py $Q > $G/v2_train_p$pr.json 2> $G/v2_train_p$pr.err
done
timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q --tb=short -p no:cacheprovider -s -k "bf16" > $G/v2_pytest.log 2>&1; echo "pytest rc=$?" >> $G/v2_pytest.log
grep -E "passed|failed|error|ratio" $G/v2_pytest.log | tail -5
python - <<'PY'
import re
rows = {}
cur = None
order = []
for ln in open('gpurun_out/v2_probe.md'):
    m = re.match(r'## DR_CONV_PRIO=(\d+)', ln)
    if m: cur = m.group(1); order.append(cur); continue
    f = [x.strip() for x in ln.strip().strip('|').split('|')]
    if len(f) == 8 and f[0].isdigit():
        rows.setdefault(tuple(f[:5]), {})[cur] = f[6]
print('shape | ' + ' | '.join(order))
for k, v in rows.items():
    print(':'.join(k) + ' | ' + ' | '.join(v.get(o, '-') for o in order))
PY
for pr in 0 6 40 72 136 73; do python -c "
import json;d=json.load(open('$G/v2_train_p$pr.json'));print('train prio=$pr',round(d['value'],1),round(d['ms_per_step'],3))" 2>/dev/null || { echo "p$pr FAILED"; tail -3 $G/v2_train_p$pr.err; }; done
