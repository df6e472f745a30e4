#!/bin/bash
# Build libvqhip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
# VQ_KEEP_TEMPS=1 additionally keeps the device assembly under build/asm/.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
OUT="$HERE/../libvqhip.so"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function
       -I"$ROOT/include" -I"$HERE")
if [[ "${VQ_KEEP_TEMPS:-0}" == "1" ]]; then
    mkdir -p "$ROOT/build/asm"
    (cd "$ROOT/build/asm" && "$HIPCC" "${FLAGS[@]}" -save-temps "$HERE/vqhip.hip" -o "$OUT")
else
    "$HIPCC" "${FLAGS[@]}" "$HERE/vqhip.hip" -o "$OUT"
fi
echo "built $OUT"
