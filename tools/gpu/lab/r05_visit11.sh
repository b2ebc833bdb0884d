#!/usr/bin/env bash
# round 5, visit 11: SQ / TCC counters of the fp32 16-column tiles on their own layers (evidence of the bound), and of the
# x3 weight-gradient kernel
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" \
      "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
      "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM")
probe() {   # name match env... -- shape
  name=$1; match=$2; shift 2
  cd /tmp; i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    env PROBE_B=200 "$@" timeout 200 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_${name}_$i -o p -- python $R/tools/conv_one.py $SHAPE > $R/gpurun_out/pmclog_${name}_$i.txt 2>&1
  done
  cd $R
  python tools/rocpd_counters.py gpurun_out/pmc_${name}_*/p_results.db --match $match > gpurun_out/v11_counters_$name.md 2> gpurun_out/v11_counters_$name.err
  rm -rf gpurun_out/pmc_${name}_*
}
SHAPE="32 78 78 3 -1 3" probe igemm16_3x3_78 conv_igemm DR_X3_BN96=0
SHAPE="32 256 156 1 -1 3" probe igemm16_1x1_256_156 conv_igemm A=1
SHAPE="32 78 78 3 -1 3" probe x3_96_3x3_78 conv_x3 A=1
head -30 gpurun_out/v11_counters_igemm16_3x3_78.md; head -12 gpurun_out/v11_counters_igemm16_1x1_256_156.md; head -12 gpurun_out/v11_counters_x3_96_3x3_78.md
# the whole GPU suite with the x3 kernels forced onto EVERY layer they can run (DR_CONV_X3=2: shallow grids, 16x16 and below, K < 128 too)
cd $R; rm -f gpurun_out/pytest_live.log
( time DR_CONV_X3=2 timeout 1100 python -m pytest tests/ -q -m gpu -p no:cacheprovider ) > gpurun_out/v11_suite_x3_everywhere.log 2>&1; echo "rc=$?" >> gpurun_out/v11_suite_x3_everywhere.log
grep -v "start\]\|passed\]" gpurun_out/v11_suite_x3_everywhere.log | tail -15
