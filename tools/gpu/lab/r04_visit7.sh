#!/usr/bin/env bash
# round 4 visit 7: where the fused hourglass bottom's 114 us go -- ablation builds (no MFMAs / no weight fetches / no barriers / no LDS
# fragment reads), HIP-event time of the launch at B = 40 and B = 1
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; G=gpurun_out
for v in "" hgabl1 hgabl2 hgabl3 hgabl4; do
for B in 40 1; do
DR_LIB_VARIANT=$v timeout 300 python bench.py --mode infer --batch $B --replicas 1 --merge 1 --no-cpu-baseline --steps 20 --warmup 5 --detail $G/r04_v7_detail_${v:-base}_b$B.md > /dev/null 2> $G/r04_v7_${v:-base}_b$B.err; echo "${v:-base} B=$B $(grep hourglass $G/r04_v7_detail_${v:-base}_b$B.md)"
done
done
