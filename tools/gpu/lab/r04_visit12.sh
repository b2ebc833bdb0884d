#!/usr/bin/env bash
# round 4 visit 12: where the bf16-stored raw outputs win and lose: kernel traces of the bf16 training step with DR_BF16_RAW=1 / 0
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; G=gpurun_out
P="--no-cpu-baseline --no-profile --no-forward-vote --precision bf16 --steps 10 --warmup 5"
cd /tmp
DR_PIPELINE=1 DR_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$G/prof_v12a -o t -- python $R/bench.py $P > $R/$G/r04_v12_a.log 2>&1
DR_BF16_RAW=0 DR_PIPELINE=1 DR_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$G/prof_v12b -o t -- python $R/bench.py $P > $R/$G/r04_v12_b.log 2>&1
cd $R
for v in a b; do
  db=$(ls $G/prof_v12$v/*_results.db 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db "bench.py bf16 (visit 12, $v: a = raw bf16, b = raw fp32)" > $G/r04_v12_kernel_stats_$v.md && rm -rf $G/prof_v12$v
done
head -30 $G/r04_v12_kernel_stats_a.md | cut -c1-150
