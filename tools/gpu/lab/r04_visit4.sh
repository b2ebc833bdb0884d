#!/usr/bin/env bash
# round 4 visit 4: the fused hourglass bottoms (hg_fused.h) -- parity tests, latency by batch size fused / unfused, forward+vote
# throughput with one engine and with the replica pool; training with the grouping threshold of visit 3
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; G=gpurun_out
timeout 600 python -m pytest tests/test_fused_tail.py tests/test_forward_parity.py tests/test_gpu_fullsize.py tests/test_pipeline.py tests/test_gpu_configs.py -m gpu -q --tb=short -p no:cacheprovider -k "not train and not rccl" > $G/r04_v4_tests.log 2>&1; echo "rc=$?" >> $G/r04_v4_tests.log
tail -6 $G/r04_v4_tests.log
timeout 600 python tools/latency_bench.py > $G/r04_v4_latency.md 2>&1; cat $G/r04_v4_latency.md
Q="--no-cpu-baseline --steps 100 --warmup 10 --no-profile"
b() { name=$1; shift; env "$@" timeout 200 python bench.py $Q --mode infer > $G/r04_v4_$name.json 2> $G/r04_v4_$name.err; python - <<PY
import json
try:
    d=json.load(open('$G/r04_v4_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3), 'single', (d['config'].get('single_replica') or {}).get('value'))
except Exception as e: print('$name failed', e)
PY
}
b pool_fused A=1
b pool_unfused DR_FUSE_TAIL=0
timeout 200 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --no-profile --no-forward-vote > $G/r04_v4_train.json 2> $G/r04_v4_train.err; python -c "import json; d=json.load(open('$G/r04_v4_train.json')); print('train', round(d['value'],1))"
timeout 300 python bench.py --mode infer --replicas 1 --merge 1 --no-cpu-baseline --steps 40 --warmup 10 --detail $G/r04_v4_detail_infer.md > $G/r04_v4_infer_prof.json 2> $G/r04_v4_infer_prof.err; head -12 $G/r04_v4_detail_infer.md; grep hourglass $G/r04_v4_detail_infer.md
