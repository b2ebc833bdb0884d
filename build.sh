#!/usr/bin/env bash
# Build the native libraries (gfx950):
#   densereg_amd/lib/libdensereg_hip.so       the product: the C ABI of include/densereg.h, nothing else
#   densereg_amd/lib/libdensereg_hip_dbg.so   the same sources + the dr_dbg_* test / micro-benchmark hooks (tests/, tools/)
# and, with --emu, the host-fiber test library.  DR_OUT_DIR overrides the output directory (tools/kernel_resources.py builds
# into a temporary directory, never over the shipped binary); --product-only skips the debug library.
set -euo pipefail
cd "$(dirname "$0")"
SRC=densereg_amd/csrc
OUT=${DR_OUT_DIR:-densereg_amd/lib}
mkdir -p "$OUT"
if [[ "${1:-}" == "--emu" ]]; then
    mkdir -p tests/hipemu/_build
    /opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O2 -g -fPIC -shared -DDR_EMU -DDR_DEBUG_HOOKS -ffp-contract=off \
        -Itests/hipemu -I$SRC -Iinclude -Wno-unused-value -Wno-unknown-pragmas \
        -x c++ $SRC/densereg.cpp tests/hipemu/hip_emu.cpp -o tests/hipemu/_build/libdensereg_emu.so.tmp -lpthread
    mv -f tests/hipemu/_build/libdensereg_emu.so.tmp tests/hipemu/_build/libdensereg_emu.so
    echo "built tests/hipemu/_build/libdensereg_emu.so"
else
    HIPCC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -shared -Iinclude -I$SRC"
    # each library is written next to its destination and renamed into place: a failed or interrupted build never leaves a
    # half-written .so where the loader (or the GPU box snapshot) would pick it up
    pids=()
    if [[ "${1:-}" != "--product-only" ]]; then
        ( $HIPCC ${DR_HIPCC_EXTRA:-} -DDR_DEBUG_HOOKS -x hip $SRC/densereg.cpp -o $OUT/libdensereg_hip_dbg.so.tmp && mv -f $OUT/libdensereg_hip_dbg.so.tmp $OUT/libdensereg_hip_dbg.so ) &
        pids+=($!)
    fi
    $HIPCC ${DR_HIPCC_EXTRA:-} -x hip $SRC/densereg.cpp -o $OUT/libdensereg_hip.so.tmp
    mv -f $OUT/libdensereg_hip.so.tmp $OUT/libdensereg_hip.so
    for p in "${pids[@]}"; do wait "$p"; done
    echo "built $OUT/libdensereg_hip.so$([[ "${1:-}" != "--product-only" ]] && echo " and $OUT/libdensereg_hip_dbg.so")"
fi
