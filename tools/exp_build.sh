#!/bin/bash
# Build timing-experiment variants of the library (GQ_PF_EXP=n) into tools/exp/.
cd "$(dirname "$0")/.."
for n in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -fvisibility=hidden -DGQ_PF_EXP=$n \
    -Iinclude -Igradient-quantization_amd/csrc -o tools/exp/libgq_exp$n.so gradient-quantization_amd/csrc/*.hip 2>&1 | grep -E " error" -A3
done
ls tools/exp
