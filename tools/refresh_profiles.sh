#!/usr/bin/env bash
# Turn the scratch output of tools/gpu/r03_record.sh (gpurun_out/) into the tracked summaries under profiles/.
#   tools/refresh_profiles.sh r03
set -e
R=${1:-r03}
cd "$(dirname "$0")/.."
G=gpurun_out
python tools/rocpd_summary.py $G/prof_train/train_results.db "bench.py --steps 10 --warmup 5 (train mode, round ${R#r}; default executor: the five micro-steps of an accumulation window as one pass of launches + the side stream of the full-resolution weight gradients, whose kernels overlap the main stream's)" > profiles/${R}_train_kernel_stats.md
[ -f $G/prof_train_inline/train_results.db ] && python tools/rocpd_summary.py $G/prof_train_inline/train_results.db "DR_PIPELINE=1 DR_WGRAD_STREAM=0 bench.py --steps 10 --warmup 5 (train mode, round ${R#r}; everything on the caller's stream: what bench.py's roofline leg times)" > profiles/${R}_train_kernel_stats_inline.md
python tools/rocpd_summary.py $G/prof_infer/infer_results.db "bench.py --mode infer --replicas 1 --steps 10 --warmup 5 (round ${R#r})" > profiles/${R}_infer_kernel_stats.md
for m in train infer train_bf16 train_bf16_s4f256hw256; do
  if [ -f $G/pmc_${m}_fetch/fetch_results.db ] && [ -f $G/pmc_${m}_write/write_results.db ]; then
    python tools/rocpd_pmc.py $G/pmc_${m}_fetch/fetch_results.db $G/pmc_${m}_write/write_results.db > profiles/${R}_pmc_traffic_${m}.md
    python tools/rocpd_pmc.py $G/pmc_${m}_fetch/fetch_results.db $G/pmc_${m}_write/write_results.db --json $m profiles/pmc_traffic.json $G/pmc_${m}_stamp.json
  fi
done
for n in train infer msra c5_bf16 c5_f32 train_bf16 torchrun allreduce train_g1 train_g1_depth1; do
  [ -s $G/${R}_bench_$n.json ] && cp $G/${R}_bench_$n.json profiles/${R}_bench_$n.json
done
cp $G/${R}_detail_train.md profiles/${R}_train_per_layer.md
cp $G/${R}_detail_infer.md profiles/${R}_infer_per_layer.md
[ -f $G/${R}_detail_c5_bf16.md ] && cp $G/${R}_detail_c5_bf16.md profiles/${R}_config5_train_per_layer_bf16.md
[ -f $G/${R}_latency.md ] && cp $G/${R}_latency.md profiles/${R}_infer_latency_by_batch.md
[ -f $G/${R}_groups_test_gpu.log ] && cp $G/${R}_groups_test_gpu.log profiles/${R}_groups_test_gpu.log
[ -f $G/test_branches.jsonl ] && cp $G/test_branches.jsonl profiles/${R}_gpu_test_branches.jsonl
echo refreshed profiles/${R}_*
