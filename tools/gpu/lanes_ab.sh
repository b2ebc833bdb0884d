#!/usr/bin/env bash
# GPU visit: A/B of executor lanes (multi-stream vs DR_SINGLE_STREAM=1) at several batch sizes
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for B in ${BATCHES:-4 16 40}; do
  for mode in train infer; do
    a=$(timeout 300 python bench.py --mode $mode --batch $B --steps 40 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
    b=$(DR_SINGLE_STREAM=1 timeout 300 python bench.py --mode $mode --batch $B --steps 40 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
    echo "$mode B=$B lanes=$a single=$b"
  done
done
