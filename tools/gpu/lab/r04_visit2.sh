#!/usr/bin/env bash
# round 4 visit 2: re-run of the two re-barred bench-shape tests; weight-gradient sweeps at 200 crops per launch (tile rule, the
# 16x16-tile kernels, slab counts); training A/B of the new defaults; upper bound of "BatchReNorm apply in the reader's loader"
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; G=gpurun_out
timeout 300 python -m pytest tests/test_bench_shapes.py -m gpu -q -s --tb=short -p no:cacheprovider -k "micro_step_loop or replica" > $G/r04_v2_bench_shapes.log 2>&1; echo "rc=$?" >> $G/r04_v2_bench_shapes.log
timeout 300 python -m pytest tests/test_train_parity.py tests/test_groups.py -m gpu -q --tb=short -p no:cacheprovider > $G/r04_v2_train_tests.log 2>&1; echo "rc=$?" >> $G/r04_v2_train_tests.log
export PROBE_B=200
W=$G/r04_v2_wgrad.md; : > $W
run() { PROBE_SHAPES=$1 PROBE_T=$2 PROBE_NS=$3 timeout 300 python tools/wgrad_bench.py | tail -n +3 >> $W; }
echo "## tile rule" >> $W
run 32:515:512:1,32:156:256:1,32:131:128:1 64,128 0,32,64,128
echo "## 16x16 tiles: kernel rows" >> $W
run 32:78:78:3,32:65:65:3 96,161 0,64,128,256,512
run 64:16:16:3,32:32:32:3 64,162 0,64,128,256,512
echo "## 16x16 tiles: 1x1" >> $W
run 32:156:78:1,32:131:65:1 64,128,164 0,32,64,128,256
run 32:78:256:1,32:65:128:1,32:70:128:1 128,163 0,32,64,128,256
run 64:32:16:1,64:16:64:1,64:32:64:1,32:64:32:1,32:32:64:1,32:32:128:1 64,165 0,64,128,256,512
echo "## slab counts, big layers" >> $W
run 32:256:256:3,32:128:128:3 128 0,16,24,32,40,48,64,96,128
run 32:512:512:1,32:256:512:1,32:512:256:1,32:128:256:1,32:256:128:1 128 0,16,32,48,64,96,128,192
run 32:64:64:3,32:128:64:1,32:64:128:1 64 0,32,64,128,192,256,384
Q="--no-cpu-baseline --no-forward-vote --steps 40 --warmup 10"
b() { name=$1; shift; env "$@" timeout 200 python bench.py $Q --no-profile > $G/r04_v2_$name.json 2> $G/r04_v2_$name.err; python - <<PY
import json
try: d=json.load(open('$G/r04_v2_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)
PY
}
b default A=1
b wg16_off DR_WG16=0
b tilerule_off DR_WG_TILE_RULE=0
b both_off DR_WG16=0 DR_WG_TILE_RULE=0
b group_maxm DR_GROUP_MAXM=65536
b skip_sr_apply DR_EXP_SKIP_SR_APPLY=2
b default2 A=1
timeout 300 python bench.py $Q --detail $G/r04_v2_detail_train.md > $G/r04_v2_train_prof.json 2> $G/r04_v2_train_prof.err
tail -6 $G/r04_v2_bench_shapes.log; tail -4 $G/r04_v2_train_tests.log; cat $W
