#!/bin/bash
# Experiment builds of libvqhip: tools/build_exp.sh <name> [-DMACRO=value ...]  ->  build/exp/libvqhip_<name>.so
# (git-ignored, but they travel to the GPU box; select one with VQHIP_LIB=build/exp/libvqhip_<name>.so)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
name=$1; shift
mkdir -p "$ROOT/build/exp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-function \
    -I"$ROOT/include" -I"$ROOT/vector_quantization_amd/csrc" "$@" "$ROOT/vector_quantization_amd/csrc/vqhip.hip" \
    -o "$ROOT/build/exp/libvqhip_$name.so"
echo "built build/exp/libvqhip_$name.so ($*)"
