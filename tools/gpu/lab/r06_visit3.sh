#!/usr/bin/env bash
# round 6, visit 3: SQ / TCC counters of conv_p3_kernel (3x3 256->256, 200 crops): the product order (0), copies between the MFMA groups (1),
# no copies in the loop (2); conv_x3_kernel beside them (PROBE_X3=2)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
SHAPE="${SHAPE:-32 256 256 3 -1 3}"
cd /tmp
for cfg in "6 0" "6 1" "6 2" "2 0"; do
  set -- $cfg; x3=$1; var=$2; i=0
  for cs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
    i=$((i+1))
    PROBE_B=200 PROBE_X3=$x3 DR_P3_VARIANT=$var timeout 300 rocprofv3 --kernel-trace --pmc $cs -d $R/gpurun_out/p3pmc_${x3}_${var}_$i -o p -- python $R/tools/conv_one.py $SHAPE > $R/gpurun_out/p3pmc_${x3}_${var}_$i.log 2>&1; echo "rc=$?" >> $R/gpurun_out/p3pmc_${x3}_${var}_$i.log
  done
  ( cd $R; echo "# PROBE_X3=$x3 DR_P3_VARIANT=$var"; python tools/rocpd_counters.py gpurun_out/p3pmc_${x3}_${var}_*/p_results.db --match conv_ ) >> $R/gpurun_out/r06v3_p3_counters.md 2>> $R/gpurun_out/r06v3_p3_counters.err
done
cd $R; cat gpurun_out/r06v3_p3_counters.md; tail -3 gpurun_out/r06v3_p3_counters.err; tail -2 gpurun_out/p3pmc_6_0_1.log
rm -rf gpurun_out/p3pmc_*
