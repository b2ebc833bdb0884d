#!/usr/bin/env bash
# Build compile-time variants of the libraries for A/B measurements on the GPU box:
#   tools/build_variants.sh name1 "-DFLAG=1 ..." name2 "-D..." ...
# -> densereg_amd/lib/variants/<name>/libdensereg_hip{,_dbg}.so, selected at run time with DR_LIB_VARIANT=<name>.
set -euo pipefail
cd "$(dirname "$0")/.."
pids=()
while [[ $# -ge 2 ]]; do
    name=$1; flags=$2; shift 2
    out=densereg_amd/lib/variants/$name
    mkdir -p "$out"
    ( DR_OUT_DIR=$out DR_HIPCC_EXTRA="$flags" ./build.sh > "$out/build.log" 2>&1 || { echo "variant $name FAILED"; tail -5 "$out/build.log"; } ) &
    pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
ls -la densereg_amd/lib/variants/*/libdensereg_hip.so
