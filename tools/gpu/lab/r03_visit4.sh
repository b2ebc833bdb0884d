#!/usr/bin/env bash
# round 3, visit 4: where do kernel arguments live?  HIP_FORCE_DEV_KERNARG=1 puts the argument blocks in device memory instead of
# host-coherent memory (every workgroup's first scalar loads); plus the bare-loop ablations 12 / 13 (epilogue stores on / off)
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
Q="--no-cpu-baseline --no-profile --no-forward-vote --steps 40 --warmup 8"
for k in default 0 1 default 1; do
  if [ $k = default ]; then timeout 200 python bench.py $Q > $G/v4_train_$k.json 2> $G/v4_train_$k.err
  else HIP_FORCE_DEV_KERNARG=$k timeout 200 python bench.py $Q > $G/v4_train_$k.json 2> $G/v4_train_$k.err; fi
  python -c "
import json;d=json.load(open('$G/v4_train_$k.json'));print('train kernarg=$k',round(d['value'],1),round(d['ms_per_step'],3))" 2>/dev/null || { echo "$k FAILED"; tail -3 $G/v4_train_$k.err; }
done
for k in default 0 1; do
  if [ $k = default ]; then timeout 200 python bench.py --mode infer $Q > $G/v4_infer_$k.json 2> $G/v4_infer_$k.err
  else HIP_FORCE_DEV_KERNARG=$k timeout 200 python bench.py --mode infer $Q > $G/v4_infer_$k.json 2> $G/v4_infer_$k.err; fi
  python -c "
import json;d=json.load(open('$G/v4_infer_$k.json'));print('infer kernarg=$k',round(d['value'],1),round(d['ms_per_step'],3))" 2>/dev/null || { echo "$k FAILED"; tail -3 $G/v4_infer_$k.err; }
done
for k in 0 1; do echo "latency kernarg=$k"; HIP_FORCE_DEV_KERNARG=$k timeout 200 python tools/latency_bench.py 2>&1 | tail -6; done
SH="32:512:512:1:1:0 32:512:512:1:1:12 32:512:512:1:1:13 32:256:256:3:1:0 32:256:256:3:1:12 32:256:256:3:1:13 32:128:128:1:3:0 64:512:512:1:1:12 64:512:512:1:1:13"
for k in 0 1; do echo "probe kernarg=$k"; HIP_FORCE_DEV_KERNARG=$k timeout 200 python tools/conv_probe.py $SH 2>&1 | tail -9; done
