#!/usr/bin/env bash
# round 3, visit 27: ReplicaPool merge on the GPU (parity test), bench infer with the new defaults, latency table
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_pipeline.py tests/test_host_mirror.py -m gpu -x -q 2>&1 | tail -5
Q="--no-cpu-baseline --steps 100 --warmup 10"
run() { name=$1; shift; env "$@" > $G/v27_$name.json 2> $G/v27_$name.err; python -c "
import json;d=json.load(open('$G/v27_$name.json'));r=d.get('roofline') or {};print('$name',round(d['value'],1),round(d['ms_per_step'],3),(d['config'].get('single_replica') or {}).get('value'), r.get('kernel'), r.get('frac'), r.get('avg_launch_us'))" 2>/dev/null || { echo "$name FAILED"; tail -5 $G/v27_$name.err; }; }
run infer timeout 300 python bench.py $Q --mode infer
run infer_r3m5 timeout 300 python bench.py $Q --mode infer --replicas 3 --no-profile
run infer_r2m8 timeout 300 python bench.py $Q --mode infer --merge 8 --no-profile
run infer_r2m4 timeout 300 python bench.py $Q --mode infer --merge 4 --no-profile
run infer_bf16 timeout 300 python bench.py $Q --mode infer --precision bf16 --no-profile
run train timeout 300 python bench.py $Q
