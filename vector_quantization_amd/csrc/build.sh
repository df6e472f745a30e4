#!/bin/bash
# Build libvqhip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
# VQ_KEEP_TEMPS=1 keeps the device assembly under build/asm/ instead — and leaves the shipped library alone: the same sources
# built with -save-temps from another directory come out with another __hip_cuid (another sha256 of the file, same code), and
# bench.py reports the PMC traffic of profiles/pmc_latest.json only for the sha256 it was collected with.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
OUT="$HERE/../libvqhip.so"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function
       -I"$ROOT/include" -I"$HERE")
if [[ "${VQ_KEEP_TEMPS:-0}" == "1" ]]; then
    mkdir -p "$ROOT/build/asm"
    OUT="$ROOT/build/asm/libvqhip_temps.so"
    (cd "$ROOT/build/asm" && "$HIPCC" "${FLAGS[@]}" -save-temps "$HERE/vqhip.hip" -o "$OUT")
else
    "$HIPCC" "${FLAGS[@]}" "$HERE/vqhip.hip" -o "$OUT"
fi
echo "built $OUT"
