#!/usr/bin/env bash
# Turn the scratch output of tools/gpu/full_visit.sh (gpurun_out/) into the tracked summaries under profiles/.
#   tools/refresh_profiles.sh r01
set -e
R=${1:-r01}
cd "$(dirname "$0")/.."
G=gpurun_out
python tools/rocpd_summary.py $G/prof_train/train_results.db "bench.py --steps 10 --warmup 5 (train mode, round ${R#r})" > profiles/${R}_train_kernel_stats.md
python tools/rocpd_summary.py $G/prof_infer/infer_results.db "bench.py --mode infer --steps 10 --warmup 5 (round ${R#r})" > profiles/${R}_infer_kernel_stats.md
python tools/rocpd_pmc.py $G/pmc_fetch/fetch_results.db $G/pmc_write/write_results.db > profiles/${R}_train_pmc_traffic.md
python tools/rocpd_pmc.py $G/pmc_fetch_infer/fetch_results.db $G/pmc_write_infer/write_results.db > profiles/${R}_infer_pmc_traffic.md
python tools/rocpd_pmc.py $G/pmc_fetch/fetch_results.db $G/pmc_write/write_results.db --json train profiles/pmc_traffic.json
python tools/rocpd_pmc.py $G/pmc_fetch_infer/fetch_results.db $G/pmc_write_infer/write_results.db --json infer profiles/pmc_traffic.json
cp $G/bench_train.json profiles/${R}_bench_train.json
cp $G/bench_infer.json profiles/${R}_bench_infer.json
cp $G/detail_train.md profiles/${R}_train_per_layer.md
cp $G/detail_infer.md profiles/${R}_infer_per_layer.md
for p in bf16; do
  [ -f $G/bench_infer_$p.json ] && cp $G/bench_infer_$p.json profiles/${R}_bench_infer_$p.json && cp $G/detail_infer_$p.md profiles/${R}_infer_per_layer_$p.md
  [ -f $G/bench_train_$p.json ] && cp $G/bench_train_$p.json profiles/${R}_bench_train_$p.json && cp $G/detail_train_$p.md profiles/${R}_train_per_layer_$p.md
done
[ -f $G/conv_bench.md ] && cp $G/conv_bench.md profiles/${R}_conv_microbench.md
[ -f $G/conv_counters.md ] && cp $G/conv_counters.md profiles/${R}_conv_sq_counters.md
echo refreshed profiles/${R}_*
