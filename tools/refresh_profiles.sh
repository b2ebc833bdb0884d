#!/usr/bin/env bash
# Turn the scratch output of tools/gpu/r02_record.sh (gpurun_out/) into the tracked summaries under profiles/.
#   tools/refresh_profiles.sh r02
set -e
R=${1:-r02}
cd "$(dirname "$0")/.."
G=gpurun_out
python tools/rocpd_summary.py $G/prof_train/train_results.db "bench.py --steps 10 --warmup 5 (train mode, round ${R#r}; default executor: the full-resolution weight gradients run on the side stream, so kernels overlap and the column sums exceed the step time)" > profiles/${R}_train_kernel_stats.md
[ -f $G/prof_train_inline/train_results.db ] && python tools/rocpd_summary.py $G/prof_train_inline/train_results.db "DR_WGRAD_STREAM=0 bench.py --steps 10 --warmup 5 (train mode, round ${R#r}; everything on the caller's stream: what bench.py's roofline leg times)" > profiles/${R}_train_kernel_stats_inline.md
python tools/rocpd_summary.py $G/prof_infer/infer_results.db "bench.py --mode infer --steps 10 --warmup 5 (round ${R#r})" > profiles/${R}_infer_kernel_stats.md
python tools/rocpd_pmc.py $G/pmc_fetch/fetch_results.db $G/pmc_write/write_results.db > profiles/${R}_train_pmc_traffic.md
python tools/rocpd_pmc.py $G/pmc_fetch_infer/fetch_results.db $G/pmc_write_infer/write_results.db > profiles/${R}_infer_pmc_traffic.md
python tools/rocpd_pmc.py $G/pmc_fetch/fetch_results.db $G/pmc_write/write_results.db --json train profiles/pmc_traffic.json
python tools/rocpd_pmc.py $G/pmc_fetch_infer/fetch_results.db $G/pmc_write_infer/write_results.db --json infer profiles/pmc_traffic.json
if [ -f $G/pmc_fetch_bf16/fetch_results.db ]; then
  python tools/rocpd_pmc.py $G/pmc_fetch_bf16/fetch_results.db $G/pmc_write_bf16/write_results.db > profiles/${R}_train_pmc_traffic_bf16.md
  python tools/rocpd_pmc.py $G/pmc_fetch_bf16/fetch_results.db $G/pmc_write_bf16/write_results.db --json train_bf16 profiles/pmc_traffic.json
fi
for n in train infer msra c5_bf16 c5_f32 train_bf16 torchrun allreduce; do
  [ -s $G/${R}_bench_$n.json ] && cp $G/${R}_bench_$n.json profiles/${R}_bench_$n.json
done
cp $G/${R}_detail_train.md profiles/${R}_train_per_layer.md
cp $G/${R}_detail_infer.md profiles/${R}_infer_per_layer.md
[ -f $G/${R}_detail_c5_bf16.md ] && cp $G/${R}_detail_c5_bf16.md profiles/${R}_config5_train_per_layer_bf16.md
echo refreshed profiles/${R}_*
