#!/usr/bin/env bash
# GPU visit: conv microbench + tile parity tests + SQ counters on one shape
./tools/gpu_run6.sh > gpurun_out/run6.log 2>&1
./tools/gpu_run7.sh > gpurun_out/run7.log 2>&1
tail -2 gpurun_out/pytest_tile.log; cat gpurun_out/conv_counters.md | grep -E "GRBM|MFMA_BUSY|WAVE_CYCLES|WAIT|BANK|IDX_ACTIVE|INSTS_VALU |INSTS_SALU"
grep -E "auto  |x[0-9]+  " gpurun_out/conv_bench.md
