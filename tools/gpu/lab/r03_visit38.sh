#!/usr/bin/env bash
# round 3, visit 38: SQ counters of the weight-gradient kernel (T = 128) on its biggest layer at 200 crops per launch
mkdir -p gpurun_out
export TMPDIR=/tmp PROBE_B=200 PROBE_SHAPES="32:256:256:3" PROBE_NS=0 PROBE_T=128
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/wgpmc_$i -o p -- python $R/tools/wgrad_bench.py > $R/gpurun_out/wgpmc_$i.log 2>&1; echo "rc=$?" >> $R/gpurun_out/wgpmc_$i.log
done
cd $R
python tools/rocpd_counters.py gpurun_out/wgpmc_*/p_results.db --match conv_wgrad_kernel > gpurun_out/v38_wgrad_counters_b200.md 2> gpurun_out/v38.err
cat gpurun_out/v38_wgrad_counters_b200.md; tail -2 gpurun_out/wgpmc_1.log
