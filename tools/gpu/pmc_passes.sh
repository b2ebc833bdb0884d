#!/usr/bin/env bash
# Stamped PMC passes (HBM traffic per kernel: FETCH_SIZE / WRITE_SIZE in SEPARATE rocprofv3 runs, kernel-trace only) of one bench
# configuration:  tools/gpu/pmc_passes.sh <name> <bench.py args...>   -> gpurun_out/pmc_<name>_{fetch,write}/ + pmc_<name>_stamp.json
# Runs inline on one stream (DR_PIPELINE=1 DR_WGRAD_STREAM=0) so that a kernel's counters are its own.  GIT_HEAD is passed in by the
# caller (the GPU box has no .git).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; G=$R/gpurun_out; name=$1; shift
export TMPDIR=/tmp
python - <<PY > $G/pmc_${name}_stamp.json
import json, sys, time
sys.path.insert(0, '$R')
from densereg_amd.buildinfo import kernel_source_hash
print(json.dumps({'kernel_source_hash': kernel_source_hash(), 'git_head': '${GIT_HEAD:-unknown}', 'date': time.strftime('%Y-%m-%d %H:%M:%S'),
                  'command': 'DR_PIPELINE=1 DR_WGRAD_STREAM=0 bench.py --steps ${PMC_STEPS:-2} --warmup ${PMC_WARMUP:-1} $*'}))
PY
cd /tmp
P="--no-cpu-baseline --no-profile --no-forward-vote --steps ${PMC_STEPS:-2} --warmup ${PMC_WARMUP:-1}"
DR_PIPELINE=1 DR_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $G/pmc_${name}_fetch -o fetch -- python $R/bench.py $P "$@" > $G/pmc_${name}_fetch.log 2>&1; echo "rc=$?" >> $G/pmc_${name}_fetch.log
DR_PIPELINE=1 DR_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $G/pmc_${name}_write -o write -- python $R/bench.py $P "$@" > $G/pmc_${name}_write.log 2>&1; echo "rc=$?" >> $G/pmc_${name}_write.log
tail -1 $G/pmc_${name}_fetch.log $G/pmc_${name}_write.log
