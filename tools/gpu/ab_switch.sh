#!/usr/bin/env bash
# GPU visit: A/B of an environment switch (default vs $AB_VAR=$AB_VAL), three repetitions each, train + infer
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do
  for mode in train infer; do
    a=$(timeout 300 python bench.py --mode $mode --steps 40 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
    b=$(env $AB_VAR=$AB_VAL timeout 300 python bench.py --mode $mode --steps 40 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
    echo "$mode rep$rep default=$a $AB_VAR=$AB_VAL: $b"
  done
done
