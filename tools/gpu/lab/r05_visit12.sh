#!/usr/bin/env bash
# round 5, visit 12: A/B on one box of the hand-written split (product) against the vector-conversion split, and of the epilogue batch of
# four rows (product) against eight, in the eight-wave x3 kernel -- compile-time variants (tools/build_variants.sh)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
Q="--steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > gpurun_out/v12_$name.json 2> gpurun_out/v12_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/v12_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)"; }
for rep in 1 2 3; do
  run product_$rep A=1
  run vecsplit_$rep DR_LIB_VARIANT=vecsplit
  run epb8_$rep DR_LIB_VARIANT=epb8
  run vecsplit_epb8_$rep DR_LIB_VARIANT=vecsplit_epb8
done
for v in "" vecsplit epb8 vecsplit_epb8; do
  echo "== variant '$v'"; DR_LIB_VARIANT=$v timeout 300 python tools/wgrad_x3_bench.py 200 2>/dev/null | sed -n 3,6p
  DR_LIB_VARIANT=$v timeout 300 python tools/x3_bench.py 200 2>/dev/null | sed -n 3,6p | cut -c1-80
done
