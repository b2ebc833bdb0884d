#!/usr/bin/env bash
# round 4 record visit.  Order matters: the PMC passes run FIRST and are folded into profiles/pmc_traffic.json on the box, so every
# bench line below is produced after the passes of its own build and quotes a stamped roofline.traffic.
#   GIT_HEAD=$(git rev-parse --short HEAD) gpurun -- "GIT_HEAD=$GIT_HEAD bash tools/gpu/r04_record.sh"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; G=gpurun_out
rm -f $G/test_branches.jsonl
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=10 > $G/r04_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $G/r04_pytest_gpu.log
timeout 400 python -m pytest tests/test_bench_shapes.py tests/test_groups.py tests/test_fused_tail.py -m gpu -q -s -p no:cacheprovider > $G/r04_bench_shape_tests_gpu.log 2>&1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $G/r04_smoke.log 2>&1; echo "smoke rc=$?" >> $G/r04_smoke.log
# ---- stamped PMC passes, folded into the traffic table bench.py reads -----------------------------------------------------
PMC_STEPS=5 PMC_WARMUP=5 bash tools/gpu/pmc_passes.sh train
bash tools/gpu/pmc_passes.sh infer --mode infer --replicas 1
PMC_STEPS=5 PMC_WARMUP=5 bash tools/gpu/pmc_passes.sh train_bf16 --precision bf16
bash tools/gpu/pmc_passes.sh train_bf16_s4f256hw256 --num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu --precision bf16
cd $R
for m in train infer train_bf16 train_bf16_s4f256hw256; do
  if [ -f $G/pmc_${m}_fetch/fetch_results.db ] && [ -f $G/pmc_${m}_write/write_results.db ]; then
    python tools/rocpd_pmc.py $G/pmc_${m}_fetch/fetch_results.db $G/pmc_${m}_write/write_results.db > $G/r04_pmc_traffic_${m}.md
    python tools/rocpd_pmc.py $G/pmc_${m}_fetch/fetch_results.db $G/pmc_${m}_write/write_results.db --json $m profiles/pmc_traffic.json $G/pmc_${m}_stamp.json
  fi
done
cp profiles/pmc_traffic.json $G/r04_pmc_traffic.json
rm -rf $G/pmc_*_fetch $G/pmc_*_write                       # (the databases are hundreds of MB: only the summaries travel back)
# ---- bench lines ------------------------------------------------------------------------------------------------------------
timeout 400 python bench.py --detail $G/r04_detail_train.md > $G/r04_bench_train.json 2> $G/r04_bench_train.err; echo "bench rc=$?" >> $G/r04_bench_train.err
timeout 300 python bench.py --mode infer --detail $G/r04_detail_infer.md > $G/r04_bench_infer.json 2> $G/r04_bench_infer.err
Q="--no-cpu-baseline --steps 40 --warmup 10"
timeout 200 python bench.py $Q --groups 1 --no-forward-vote --no-profile > $G/r04_bench_train_g1.json 2> $G/r04_bench_train_g1.err      # one micro-step per pass, two in flight
timeout 200 python bench.py --dataset msra $Q > $G/r04_bench_msra.json 2> $G/r04_bench_msra.err
timeout 200 python bench.py --precision bf16 $Q > $G/r04_bench_train_bf16.json 2> $G/r04_bench_train_bf16.err
C5="--num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu --no-cpu-baseline --steps 20 --warmup 5"
timeout 400 python bench.py $C5 --precision bf16 --detail $G/r04_detail_c5_bf16.md > $G/r04_bench_c5_bf16.json 2> $G/r04_bench_c5_bf16.err
timeout 400 python bench.py $C5 --no-forward-vote > $G/r04_bench_c5_f32.json 2> $G/r04_bench_c5_f32.err
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-forward-vote > $G/r04_bench_torchrun.json 2> $G/r04_bench_torchrun.err; echo "torchrun rc=$?" >> $G/r04_bench_torchrun.err
DR_FORCE_ALLREDUCE=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-forward-vote > $G/r04_bench_allreduce.json 2> $G/r04_bench_allreduce.err; echo "allreduce rc=$?" >> $G/r04_bench_allreduce.err
timeout 60 python bench.py --gpus 2 --steps 2 > $G/r04_bench_gpus2.json 2> $G/r04_bench_gpus2.err; echo "gpus2 rc=$?" >> $G/r04_bench_gpus2.err
timeout 400 python tools/latency_bench.py > $G/r04_latency.md 2>&1
# ---- rocprof kernel stats ---------------------------------------------------------------------------------------------------
P="--no-cpu-baseline --no-profile --no-forward-vote"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$G/prof_train -o train -- python $R/bench.py --steps 10 --warmup 5 $P > $R/$G/rocprof_train.log 2>&1
DR_PIPELINE=1 DR_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$G/prof_train_inline -o train -- python $R/bench.py --steps 10 --warmup 5 $P > $R/$G/rocprof_train_inline.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$G/prof_infer -o infer -- python $R/bench.py --mode infer --replicas 1 --steps 10 --warmup 5 $P > $R/$G/rocprof_infer.log 2>&1
DR_PIPELINE=1 DR_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$G/prof_train_bf16_inline -o train -- python $R/bench.py --precision bf16 --steps 10 --warmup 5 $P > $R/$G/rocprof_train_bf16_inline.log 2>&1
cd $R
for n in train train_inline infer train_bf16_inline; do
  db=$(ls $G/prof_$n/*_results.db 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db "bench.py (round 4, $n)" > $G/r04_${n}_kernel_stats.md && rm -rf $G/prof_$n
done
tail -6 $G/r04_pytest_gpu.log; tail -2 $G/r04_smoke.log
for f in train infer train_g1 msra train_bf16 c5_bf16 c5_f32 torchrun allreduce; do python - <<PY
import json
try:
    d=json.load(open('$G/r04_bench_$f.json')); fv=d.get('forward_vote') or {}
    print('$f', round(d['value'],1), d['unit'], round(d['ms_per_step'],3), 'ms', d['dtype'], '| fwd+vote', fv.get('value') and round(fv['value'],1), '| cpu', (d.get('cpu_baseline') or {}).get('value'), '| roof', (d.get('roofline') or {}).get('frac'), '| traffic', (d.get('roofline') or {}).get('traffic'))
except Exception as e:
    print('$f', 'failed', e, open('$G/r04_bench_$f.err').read()[-300:])
PY
done
tail -2 $G/r04_bench_gpus2.err; cat $G/r04_latency.md | tail -7
