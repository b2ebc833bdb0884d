#!/usr/bin/env bash
# round 4 visit 10: BatchReNorm finalize launches as one workgroup per channel (groups x split waves): tests, A/B against the
# build before it (variant "base"), rows-per-wave threshold, kernel stats of the finalize kernels
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; G=gpurun_out
timeout 600 python -m pytest tests/test_bn_layer.py tests/test_groups.py tests/test_bench_shapes.py tests/test_pipeline.py -m gpu -q --tb=short -p no:cacheprovider -k "not replica" > $G/r04_v10_tests.log 2>&1; echo "rc=$?" >> $G/r04_v10_tests.log; tail -3 $G/r04_v10_tests.log
Q="--no-cpu-baseline --no-forward-vote --steps 40 --warmup 10 --no-profile"
b() { name=$1; shift; env "$@" timeout 200 python bench.py $Q > $G/r04_v10_$name.json 2> $G/r04_v10_$name.err; python -c "import json; d=json.load(open('$G/r04_v10_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))"; }
b new A=1
b base DR_LIB_VARIANT=base
b new2 A=1
b base2 DR_LIB_VARIANT=base
b rows128 DR_BN_FIN_ROWS=128
b rows4096 DR_BN_FIN_ROWS=4096
C5="--num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu --no-cpu-baseline --steps 10 --warmup 3 --precision bf16 --no-forward-vote --no-profile"
timeout 300 python bench.py $C5 > $G/r04_v10_c5_new.json 2> $G/r04_v10_c5_new.err; python -c "import json; d=json.load(open('$G/r04_v10_c5_new.json')); print('c5_new', round(d['value'],1), round(d['ms_per_step'],3))"
DR_LIB_VARIANT=base timeout 300 python bench.py $C5 > $G/r04_v10_c5_base.json 2> $G/r04_v10_c5_base.err; python -c "import json; d=json.load(open('$G/r04_v10_c5_base.json')); print('c5_base', round(d['value'],1), round(d['ms_per_step'],3))"
P="--no-cpu-baseline --no-profile --no-forward-vote"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$G/prof_v10 -o train -- python $R/bench.py --steps 10 --warmup 5 $P > $R/$G/r04_v10_rocprof.log 2>&1
cd $R
db=$(ls $G/prof_v10/*_results.db 2>/dev/null | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py $db "bench.py (round 4 visit 10, train)" > $G/r04_v10_kernel_stats.md && rm -rf $G/prof_v10
grep -i "finalize\|bn_" $G/r04_v10_kernel_stats.md
