#!/usr/bin/env bash
# round 5, visit 9: 96-column x3 tile and eight-wave weight-gradient x3: tests, micro-benchmarks, A/B on the training step; SQ counters
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 300 python -m pytest tests/test_forward_parity.py tests/test_train_parity.py -q -m gpu -k "x3 or wgrad" -p no:cacheprovider > gpurun_out/v9_tests.log 2>&1; echo "rc=$?" >> gpurun_out/v9_tests.log
timeout 600 python tools/x3_bench.py 200 > gpurun_out/v9_x3_bench_b200.md 2> gpurun_out/v9_x3_bench_b200.err
timeout 300 python tools/wgrad_x3_bench.py 200 > gpurun_out/v9_wgrad_x3_bench.md 2> gpurun_out/v9_wgrad_x3_bench.err
DR_WGRAD_X3_W8=0 timeout 300 python tools/wgrad_x3_bench.py 200 > gpurun_out/v9_wgrad_x3_bench_w4.md 2> gpurun_out/v9_wgrad_x3_bench_w4.err
Q="--steps 10 --warmup 5 --no-cpu-baseline --no-forward-vote --no-profile"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > gpurun_out/v9_$name.json 2> gpurun_out/v9_$name.err; python -c "
import json
try:
    d=json.load(open('gpurun_out/v9_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))
except Exception as e: print('$name failed', e)"; }
run new A=1
run no96 DR_X3_BN96=0
run now8 DR_WGRAD_X3_W8=0
run old DR_X3_BN96=0 DR_WGRAD_X3_W8=0
run new2 A=1
run old2 DR_X3_BN96=0 DR_WGRAD_X3_W8=0
timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_trained_parity.py -q -m gpu -p no:cacheprovider > gpurun_out/v9_parity.log 2>&1; echo "rc=$?" >> gpurun_out/v9_parity.log
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  PROBE_B=200 timeout 200 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/x3pmc_$i -o p -- python $R/tools/conv_one.py 32 256 256 3 -1 3 > $R/gpurun_out/x3pmclog_$i.txt 2>&1; echo "rc=$?" >> $R/gpurun_out/x3pmclog_$i.txt
done
cd $R
python tools/rocpd_counters.py gpurun_out/x3pmc_*/p_results.db --match conv_x3 > gpurun_out/v9_conv_x3_sq_counters.md 2> gpurun_out/v9_conv_x3_sq_counters.err
rm -rf gpurun_out/x3pmc_*
tail -3 gpurun_out/v9_tests.log; grep "78\|65\|156\|131\|shape" gpurun_out/v9_x3_bench_b200.md; cat gpurun_out/v9_wgrad_x3_bench.md | head -9; echo w4; cat gpurun_out/v9_wgrad_x3_bench_w4.md | head -9; grep -v "start\]\|passed\]" gpurun_out/v9_parity.log | tail -3; head -32 gpurun_out/v9_conv_x3_sq_counters.md; tail -2 gpurun_out/x3pmclog_1.txt
