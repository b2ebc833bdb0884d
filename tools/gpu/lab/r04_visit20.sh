#!/usr/bin/env bash
# round 4 visit 20: kernel trace of the B=1 forward + vote (one engine, one crop per launch) and of B=40: where a millisecond goes
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; G=gpurun_out
P="--mode infer --steps 50 --warmup 10 --no-cpu-baseline --no-profile --replicas 1 --merge 1"
cd /tmp
for B in 1 40; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$G/prof_v20_$B -o t -- python $R/bench.py $P --batch $B > $R/$G/r04_v20_b$B.log 2>&1
done
cd $R
for B in 1 40; do
  db=$(ls $G/prof_v20_$B/*_results.db 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db "bench.py --mode infer --batch $B, one engine (visit 20)" > $G/r04_v20_kernel_stats_b$B.md && rm -rf $G/prof_v20_$B
done
head -40 $G/r04_v20_kernel_stats_b1.md | cut -c1-150
