#!/usr/bin/env bash
# round 4 visit 14: bf16 path: dOut of single-conv-reader activations stored as bf16 (DR_BF16_GACT): tests, A/B, kernel trace
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; G=gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "bf16 or bn_layer or config5" > $G/r04_v14_tests.log 2>&1; echo "rc=$?" >> $G/r04_v14_tests.log; tail -3 $G/r04_v14_tests.log
b() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > $G/r04_v14_$name.json 2> $G/r04_v14_$name.err; python -c "import json; d=json.load(open('$G/r04_v14_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))"; }
Q="--no-cpu-baseline --no-forward-vote --steps 40 --warmup 10 --no-profile --precision bf16"
b bf16_raw16 A=1
b bf16_gact32 DR_BF16_GACT=0
b bf16_raw16_2 A=1
b bf16_gact32_2 DR_BF16_GACT=0
Q="--num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu --no-cpu-baseline --steps 10 --warmup 3 --precision bf16 --no-forward-vote --no-profile"
b c5_raw16 A=1
b c5_gact32 DR_BF16_GACT=0
P="--no-cpu-baseline --no-profile --no-forward-vote --precision bf16 --steps 10 --warmup 5"
cd /tmp
DR_PIPELINE=1 DR_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$G/prof_v14a -o t -- python $R/bench.py $P > $R/$G/r04_v14_a.log 2>&1
cd $R
db=$(ls $G/prof_v14a/*_results.db 2>/dev/null | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py $db "bench.py bf16 (visit 14, raw bf16)" > $G/r04_v14_kernel_stats_a.md && rm -rf $G/prof_v14a
grep "bn_" $G/r04_v14_kernel_stats_a.md | cut -c1-150
