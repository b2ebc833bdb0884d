#!/usr/bin/env bash
# visit 32: rows per thread and batch in the BatchReNorm streaming kernels (DR_BN_ROWS, a build-time constant) x workgroup cap
mkdir -p gpurun_out; G=gpurun_out
Q="--no-cpu-baseline --no-profile --no-forward-vote --steps 40 --warmup 8"
for rows in 8 12 2 4; do
  DR_HIPCC_EXTRA="-DDR_BN_ROWS=$rows" ./build.sh > $G/v32_build_$rows.log 2>&1 || { tail -3 $G/v32_build_$rows.log; continue; }
  for grid in 512 1024 256; do
    DR_BN_GRID=$grid timeout 120 python bench.py $Q > $G/v32_r${rows}_g$grid.json 2> $G/v32_r${rows}_g$grid.err
    python -c "
import json;d=json.load(open('$G/v32_r${rows}_g$grid.json'));print('rows $rows grid $grid',round(d['value'],1),round(d['ms_per_step'],3))" || tail -2 $G/v32_r${rows}_g$grid.err
  done
done
