#!/bin/bash
# builds merkurio_amd/lib/libmerkurio_hip_abl<N>.so (-DMK_ABLATE=N, see scan_kernel.hip) for the
# breakdown scripts; run in the repository root before gpurun (hipcc cross-compiles without a GPU)
# usage: tools/build_ablations.sh [N ...]      default: 1 7 8 16 32 64
set -e
cd "$(dirname "$0")/.."
for a in ${@:-1 7 8 16 32 64}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -DMK_ABLATE=$a -shared \
    -o merkurio_amd/lib/libmerkurio_hip_abl$a.so merkurio_amd/csrc/scan_kernel.hip merkurio_amd/csrc/matcher.cpp \
    merkurio_amd/csrc/host_patterns.cpp merkurio_amd/csrc/host_loops.cpp &
done
wait
ls -la merkurio_amd/lib/libmerkurio_hip_abl*.so
