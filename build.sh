#!/usr/bin/env bash
# Build libdensereg_hip.so (product, gfx950) and, with --emu, the host-fiber test library.
set -euo pipefail
cd "$(dirname "$0")"
SRC=densereg_amd/csrc
OUT=densereg_amd/lib
mkdir -p "$OUT"
if [[ "${1:-}" == "--emu" ]]; then
    mkdir -p tests/hipemu/_build
    /opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O2 -g -fPIC -shared -DDR_EMU -ffp-contract=off \
        -Itests/hipemu -I$SRC -Iinclude -Wno-unused-value -Wno-unknown-pragmas \
        -x c++ $SRC/densereg.cpp tests/hipemu/hip_emu.cpp -o tests/hipemu/_build/libdensereg_emu.so -lpthread
    echo "built tests/hipemu/_build/libdensereg_emu.so"
else
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -shared -Iinclude -I$SRC \
        -x hip $SRC/densereg.cpp -o $OUT/libdensereg_hip.so ${DR_HIPCC_EXTRA:-}
    echo "built $OUT/libdensereg_hip.so"
fi
