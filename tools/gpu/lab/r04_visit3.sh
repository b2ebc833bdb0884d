#!/usr/bin/env bash
# round 4 visit 3: re-barred tests, the wgrad defaults after visit 2 (16x16 tiles without the 64x64-block variant, round-1 tile rule),
# in-step A/B of the measured slab table and of grouping the <= 16x16 layers at 200 crops
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; G=gpurun_out
timeout 300 python -m pytest tests/test_bench_shapes.py -m gpu -q -s --tb=short -p no:cacheprovider -k "micro_step_loop or replica" > $G/r04_v3_bench_shapes.log 2>&1; echo "rc=$?" >> $G/r04_v3_bench_shapes.log
timeout 300 python -m pytest tests/test_train_parity.py tests/test_groups.py -m gpu -q --tb=short -p no:cacheprovider > $G/r04_v3_train_tests.log 2>&1; echo "rc=$?" >> $G/r04_v3_train_tests.log
Q="--no-cpu-baseline --no-forward-vote --steps 40 --warmup 10"
b() { name=$1; shift; env "$@" timeout 200 python bench.py $Q --no-profile > $G/r04_v3_$name.json 2> $G/r04_v3_$name.err; python - <<PY
import json
try: d=json.load(open('$G/r04_v3_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)
PY
}
bm() { name=$1; shift; env "$@" timeout 200 python bench.py $Q --no-profile --dataset msra > $G/r04_v3_$name.json 2> $G/r04_v3_$name.err; python -c "import json; d=json.load(open('$G/r04_v3_$name.json')); print('$name', round(d['value'],1))"; }
b default A=1
b ns_table_off DR_WG_NS_TABLE=0
b group_maxm DR_GROUP_MAXM=65536
b group_maxm_nstab_off DR_GROUP_MAXM=65536 DR_WG_NS_TABLE=0
b wg16_3 DR_WG16=3
b wg16_0 DR_WG16=0
b default2 A=1
bm msra A=1
bm msra_wg16_0 DR_WG16=0
tail -5 $G/r04_v3_bench_shapes.log; tail -4 $G/r04_v3_train_tests.log
