#!/usr/bin/env bash
# round 4 visit 8: the lanes test after pinning both handles to the unfused launches; SQ counters of the two new kernels
# (conv_wgrad16 kernel-row variant on 3x3 78->78 at 200 crops; hg_tail_eval_kernel at 40 crops), separate PMC passes
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; G=gpurun_out
timeout 300 python -m pytest tests/test_train_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "lanes" > $G/r04_v8_lanes.log 2>&1; echo "rc=$?" >> $G/r04_v8_lanes.log; tail -2 $G/r04_v8_lanes.log
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  PROBE_B=200 PROBE_SHAPES=32:78:78:3 PROBE_T=161 PROBE_NS=0 timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/$G/wg16pmc_$i -o p -- python $R/tools/wgrad_bench.py > $R/$G/wg16pmc_$i.log 2>&1; echo "rc=$?" >> $R/$G/wg16pmc_$i.log
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/$G/hgpmc_$i -o p -- python $R/bench.py --mode infer --replicas 1 --merge 1 --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $R/$G/hgpmc_$i.log 2>&1; echo "rc=$?" >> $R/$G/hgpmc_$i.log
done
cd $R
python tools/rocpd_counters.py $G/wg16pmc_*/p_results.db --match conv_wgrad16 > $G/r04_v8_wgrad16_counters.md 2> $G/r04_v8_wgrad16_counters.err
python tools/rocpd_counters.py $G/hgpmc_*/p_results.db --match hg_tail > $G/r04_v8_hg_counters.md 2> $G/r04_v8_hg_counters.err
rm -rf $G/wg16pmc_* $G/hgpmc_*
cat $G/r04_v8_wgrad16_counters.md $G/r04_v8_hg_counters.md
