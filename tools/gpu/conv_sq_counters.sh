#!/usr/bin/env bash
# GPU visit: SQ counters of the conv kernel on one shape (3x3 256->256 @32, 64x128 tile), three PMC passes
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
SHAPE="${SHAPE:-32 256 256 3 1 3}"
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/convpmc_$i -o p -- python $R/tools/conv_one.py $SHAPE > $R/gpurun_out/convpmc_$i.log 2>&1; echo "rc=$?" >> $R/gpurun_out/convpmc_$i.log
done
cd $R
python tools/rocpd_counters.py gpurun_out/convpmc_*/p_results.db --match conv_igemm > gpurun_out/conv_counters.md 2> gpurun_out/conv_counters.err
cat gpurun_out/conv_counters.md; tail -3 gpurun_out/convpmc_1.log; tail -3 gpurun_out/conv_counters.err
