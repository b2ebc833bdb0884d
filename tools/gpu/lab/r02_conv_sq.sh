#!/usr/bin/env bash
# round 2: SQ counters of the default conv kernel (LDS-DMA refill, separate LDS stage objects) on 3x3 256->256 @32, four PMC passes
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
rm -rf $R/gpurun_out/convpmc_*
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/convpmc_$i -o p -- python $R/tools/conv_one.py 32 256 256 3 1 3 > $R/gpurun_out/convpmc_$i.log 2>&1; echo "rc=$?" >> $R/gpurun_out/convpmc_$i.log
done
cd $R
python tools/rocpd_counters.py gpurun_out/convpmc_*/p_results.db --match conv_igemm > gpurun_out/conv_counters.md 2> gpurun_out/conv_counters.err
cat gpurun_out/conv_counters.md; tail -2 gpurun_out/convpmc_1.log
