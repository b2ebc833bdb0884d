#!/usr/bin/env bash
# Turn the scratch output of tools/gpu/r04_record.sh (gpurun_out/) into the tracked summaries under profiles/.
#   tools/refresh_profiles.sh r04
# (The record script folds its PMC passes into profiles/pmc_traffic.json ON THE BOX, before the bench lines that quote them, and
# ships the table back as gpurun_out/r04_pmc_traffic.json: here it is only copied into place.)
set -e
R=${1:-r04}
cd "$(dirname "$0")/.."
G=gpurun_out
cp $G/${R}_pmc_traffic.json profiles/pmc_traffic.json
for m in train infer train_bf16 train_bf16_s4f256hw256; do
  [ -s $G/${R}_pmc_traffic_$m.md ] && cp $G/${R}_pmc_traffic_$m.md profiles/${R}_pmc_traffic_$m.md
done
cp $G/${R}_train_kernel_stats.md profiles/${R}_train_kernel_stats.md
cp $G/${R}_train_inline_kernel_stats.md profiles/${R}_train_kernel_stats_inline.md
cp $G/${R}_infer_kernel_stats.md profiles/${R}_infer_kernel_stats.md
[ -s $G/${R}_train_bf16_inline_kernel_stats.md ] && cp $G/${R}_train_bf16_inline_kernel_stats.md profiles/${R}_train_kernel_stats_bf16_inline.md
for n in train infer msra c5_bf16 c5_f32 train_bf16 torchrun allreduce train_g1; do
  [ -s $G/${R}_bench_$n.json ] && cp $G/${R}_bench_$n.json profiles/${R}_bench_$n.json
done
cp $G/${R}_detail_train.md profiles/${R}_train_per_layer.md
cp $G/${R}_detail_infer.md profiles/${R}_infer_per_layer.md
[ -f $G/${R}_detail_c5_bf16.md ] && cp $G/${R}_detail_c5_bf16.md profiles/${R}_config5_train_per_layer_bf16.md
[ -f $G/${R}_latency.md ] && cp $G/${R}_latency.md profiles/${R}_infer_latency_by_batch.md
[ -f $G/${R}_bench_shape_tests_gpu.log ] && cp $G/${R}_bench_shape_tests_gpu.log profiles/${R}_bench_shape_tests_gpu.log
[ -f $G/test_branches.jsonl ] && cp $G/test_branches.jsonl profiles/${R}_gpu_test_branches.jsonl
python tools/kernel_resources.py > profiles/${R}_kernel_resources.md 2>/dev/null || true
echo refreshed profiles/${R}_*
