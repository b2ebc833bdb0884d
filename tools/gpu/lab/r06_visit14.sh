#!/usr/bin/env bash
# round 6, visit 14: where the device idles inside the timed window (kernel trace of the default executor, gaps between kernels)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp
P="--no-cpu-baseline --no-forward-vote --no-profile"
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_gaps -o t -- python $R/bench.py --steps 20 --warmup 5 $P > $R/gpurun_out/r06v14_rocprof.log 2>&1
db=$(find /tmp/prof_gaps -name "*.db" | head -1)
python $R/tools/rocpd_gaps.py $db 0.5 > $R/gpurun_out/r06v14_gaps.md
DR_PIPELINE=1 DR_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_gaps2 -o t -- python $R/bench.py --steps 20 --warmup 5 $P > $R/gpurun_out/r06v14_rocprof_inline.log 2>&1
db=$(find /tmp/prof_gaps2 -name "*.db" | head -1)
python $R/tools/rocpd_gaps.py $db 0.5 > $R/gpurun_out/r06v14_gaps_inline.md
tail -3 $R/gpurun_out/r06v14_rocprof.log; head -12 $R/gpurun_out/r06v14_gaps.md; head -12 $R/gpurun_out/r06v14_gaps_inline.md
