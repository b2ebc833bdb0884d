#!/usr/bin/env bash
# round 4 visit 1: the new bench-shape parity tests + the whole GPU suite on this build (ADVICE fixes in the pipeline entry points),
# smoke, one default bench line (forward_vote with its own step count and the latency figures)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; G=gpurun_out
rm -f $G/test_branches.jsonl
timeout 900 python -m pytest tests/test_bench_shapes.py -m gpu -q -s --tb=short -p no:cacheprovider --durations=8 > $G/r04_v1_bench_shapes.log 2>&1; echo "rc=$?" >> $G/r04_v1_bench_shapes.log
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=12 --deselect tests/test_bench_shapes.py > $G/r04_v1_pytest_gpu.log 2>&1; echo "rc=$?" >> $G/r04_v1_pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $G/r04_v1_smoke.log 2>&1; echo "smoke rc=$?" >> $G/r04_v1_smoke.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $G/r04_v1_bench.json 2> $G/r04_v1_bench.err; echo "bench rc=$?" >> $G/r04_v1_bench.err
tail -25 $G/r04_v1_bench_shapes.log; tail -12 $G/r04_v1_pytest_gpu.log; tail -2 $G/r04_v1_smoke.log
python - <<PY
import json
d=json.load(open('$G/r04_v1_bench.json')); fv=d['forward_vote']
print('train', round(d['value'],1), round(d['ms_per_step'],3), 'roof', d['roofline']['frac'])
print('fwd+vote', round(fv['value'],1), 'steps', fv['steps'], 'single', round(fv['single_replica']['value'],1), 'latency', fv['latency_ms_unloaded'])
PY
