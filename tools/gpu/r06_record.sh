#!/usr/bin/env bash
# round 6 record visit.  Order matters: the PMC passes run FIRST and are folded into profiles/pmc_traffic.json on the box, so every
# bench line below is produced after the passes of its own build and quotes a stamped roofline.traffic.
#   GIT_HEAD=$(git rev-parse --short HEAD) gpurun -- "GIT_HEAD=$GIT_HEAD bash tools/gpu/r06_record.sh"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; G=gpurun_out
rm -f $G/test_branches.jsonl $G/pytest_live.log
# ---- the driver's own command, then the tests that print measurements with -s ----------------------------------------------------
( time timeout 1100 python -m pytest tests/ -x -q -m gpu ) > $G/r06_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $G/r06_pytest_gpu.log
cp $G/pytest_live.log $G/r06_pytest_live.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $G/r06_smoke.log 2>&1; echo "smoke rc=$?" >> $G/r06_smoke.log
timeout 1200 python -m pytest tests/test_trained_parity.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_bench_shapes.py tests/test_train_parity.py tests/test_groups.py tests/test_fused_tail.py -q -m gpu -s -p no:cacheprovider 2>&1 | grep -v "start\]" > $G/r06_measured_tests_gpu.log
# ---- stamped PMC passes, folded into the traffic table bench.py reads ------------------------------------------------------------
PMC_STEPS=5 PMC_WARMUP=5 bash tools/gpu/pmc_passes.sh train
bash tools/gpu/pmc_passes.sh infer --mode infer --replicas 1
bash tools/gpu/pmc_passes.sh train_bf16_s4f256hw256 --num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu --precision bf16
cd $R
for m in train infer train_bf16_s4f256hw256; do
  if [ -f $G/pmc_${m}_fetch/fetch_results.db ] && [ -f $G/pmc_${m}_write/write_results.db ]; then
    python tools/rocpd_pmc.py $G/pmc_${m}_fetch/fetch_results.db $G/pmc_${m}_write/write_results.db > $G/r06_pmc_traffic_${m}.md
    python tools/rocpd_pmc.py $G/pmc_${m}_fetch/fetch_results.db $G/pmc_${m}_write/write_results.db --json $m profiles/pmc_traffic.json $G/pmc_${m}_stamp.json
  fi
done
cp profiles/pmc_traffic.json $G/r06_pmc_traffic.json
rm -rf $G/pmc_*_fetch $G/pmc_*_write                       # (the databases are hundreds of MB: only the summaries travel back)
# ---- bench lines -------------------------------------------------------------------------------------------------------------------
timeout 400 python bench.py --detail $G/r06_detail_train.md > $G/r06_bench_train.json 2> $G/r06_bench_train.err; echo "bench rc=$?" >> $G/r06_bench_train.err
timeout 300 python bench.py --mode infer --detail $G/r06_detail_infer.md > $G/r06_bench_infer.json 2> $G/r06_bench_infer.err
timeout 300 python bench.py --mode infer --replicas 1 --merge 1 --steps 100 --warmup 10 --no-cpu-baseline --detail $G/r06_detail_infer_b40.md > $G/r06_bench_infer_b40.json 2> $G/r06_bench_infer_b40.err   # one engine, one 40-crop batch per launch: config 2 as stated
Q="--no-cpu-baseline --steps 40 --warmup 10"
DR_CONV_X3=0 timeout 200 python bench.py $Q --no-forward-vote --detail $G/r06_detail_train_x3off.md > $G/r06_bench_train_x3off.json 2> $G/r06_bench_train_x3off.err   # the fp32 matrix cores only (round 4's kernels), same box
DR_X3_HALO=0 DR_X3_BD=0 DR_WG_TAIL=0 DR_X3_BN160=0 timeout 200 python bench.py $Q --no-forward-vote --detail $G/r06_detail_train_r05kernels.md > $G/r06_bench_train_r05kernels.json 2> $G/r06_bench_train_r05kernels.err   # round 5's kernel selection (no halo kernel, register-staged weights, no tail split, no 160-column tile; the BatchReNorm passes keep their non-temporal hints: a compile-time switch), same box
timeout 200 python bench.py $Q --groups 1 --no-forward-vote --no-profile > $G/r06_bench_train_g1.json 2> $G/r06_bench_train_g1.err      # one micro-step per pass, two in flight
timeout 200 python bench.py --dataset msra $Q > $G/r06_bench_msra.json 2> $G/r06_bench_msra.err
timeout 200 python bench.py --precision bf16 $Q > $G/r06_bench_train_bf16.json 2> $G/r06_bench_train_bf16.err
C5="--num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu --no-cpu-baseline --steps 20 --warmup 5"
timeout 400 python bench.py $C5 --precision bf16 --detail $G/r06_detail_c5_bf16.md > $G/r06_bench_c5_bf16.json 2> $G/r06_bench_c5_bf16.err
timeout 400 python bench.py $C5 --no-forward-vote --detail $G/r06_detail_c5_f32.md > $G/r06_bench_c5_f32.json 2> $G/r06_bench_c5_f32.err
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-forward-vote > $G/r06_bench_torchrun.json 2> $G/r06_bench_torchrun.err; echo "torchrun rc=$?" >> $G/r06_bench_torchrun.err
DR_FORCE_ALLREDUCE=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-forward-vote > $G/r06_bench_allreduce.json 2> $G/r06_bench_allreduce.err; echo "allreduce rc=$?" >> $G/r06_bench_allreduce.err
timeout 60 python bench.py --gpus 2 --steps 2 > $G/r06_bench_gpus2.json 2> $G/r06_bench_gpus2.err; echo "gpus2 rc=$?" >> $G/r06_bench_gpus2.err
timeout 400 python tools/latency_bench.py > $G/r06_latency.md 2>&1
# ---- rocprof kernel stats ------------------------------------------------------------------------------------------------------------
P="--no-cpu-baseline --no-profile --no-forward-vote"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$G/prof_train -o train -- python $R/bench.py --steps 10 --warmup 5 $P > $R/$G/rocprof_train.log 2>&1
DR_PIPELINE=1 DR_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$G/prof_train_inline -o train -- python $R/bench.py --steps 10 --warmup 5 $P > $R/$G/rocprof_train_inline.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$G/prof_infer -o infer -- python $R/bench.py --mode infer --replicas 1 --steps 10 --warmup 5 $P > $R/$G/rocprof_infer.log 2>&1
cd $R
for n in train train_inline infer; do
  db=$(ls $G/prof_$n/*_results.db 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db "bench.py (round 6, $n)" > $G/r06_${n}_kernel_stats.md && python tools/rocpd_gaps.py $db 0.5 > $G/r06_${n}_gaps.md && rm -rf $G/prof_$n
done
# ---- SQ counters of the dominant kernels at 200 crops: the halo kernel (3x3 256->256), conv_x3_kernel on the same layer (PROBE_X3=7) and on 1x1 512->512
cd /tmp
for cfg in "halo 2 32 256 256 3" "x3_3x3 7 32 256 256 3" "x3_1x1 2 32 512 512 1"; do
  set -- $cfg; tag=$1; x3=$2; shift 2; i=0
  for set_ in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" \
              "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
              "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    PROBE_B=200 PROBE_X3=$x3 timeout 200 rocprofv3 --kernel-trace --pmc $set_ -d $R/$G/x3pmc_${tag}_$i -o p -- python $R/tools/conv_one.py $* -1 3 > $R/$G/x3pmc_${tag}_$i.log 2>&1; echo "rc=$?" >> $R/$G/x3pmc_${tag}_$i.log
  done
  ( cd $R; echo "# $tag: PROBE_X3=$x3 conv_one.py $* (200 crops)"; python tools/rocpd_counters.py $G/x3pmc_${tag}_*/p_results.db --match conv_x3 ) >> $R/$G/r06_conv_x3_sq_counters.md 2>> $R/$G/r06_conv_x3_sq_counters.err
done
cd $R
rm -rf $G/x3pmc_*
timeout 300 python tools/x3h_bench.py 200 > $G/r06_x3h_microbench.md 2>/dev/null
timeout 300 python tools/x3h_rule_bench.py > $G/r06_x3h_rule.md 2>/dev/null
timeout 300 python tools/p3_bench.py 200 > $G/r06_p3_microbench.md 2>/dev/null
timeout 200 python tools/x3_intercept_bench.py 200 > $G/r06_x3_intercept.md 2>/dev/null
cd /tmp; PYTHONPATH=$R timeout 600 python -m densereg_amd.model.hourglass_um_crop_tiny --dataset nyu --num_stack 2 --num_fea 128 --is_train True --max_steps 90 --synthetic_crops 2000 2>&1 | grep "^\[train\]" > $R/$G/r06_cli_train.log; cd $R
tail -6 $G/r06_pytest_gpu.log; tail -2 $G/r06_smoke.log
for f in train infer infer_b40 train_x3off train_r05kernels train_g1 msra train_bf16 c5_bf16 c5_f32 torchrun allreduce; do python - <<PY
import json
try:
    d=json.load(open('$G/r06_bench_$f.json')); fv=d.get('forward_vote') or {}
    print('$f', round(d['value'],1), d['unit'], round(d['ms_per_step'],3), 'ms', d['dtype'], '| fwd+vote', fv.get('value') and round(fv['value'],1), '| cpu', (d.get('cpu_baseline') or {}).get('value'), '| roof', (d.get('roofline') or {}).get('kernel'), (d.get('roofline') or {}).get('frac'), '| traffic', (d.get('roofline') or {}).get('traffic'))
except Exception as e:
    print('$f', 'failed', e, open('$G/r06_bench_$f.err').read()[-300:])
PY
done
tail -2 $G/r06_bench_gpus2.err; tail -7 $G/r06_latency.md; head -30 $G/r06_conv_x3_sq_counters.md; cat $G/r06_cli_train.log
