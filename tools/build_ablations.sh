#!/bin/bash
# builds merkurio_amd/lib/libmerkurio_hip_abl<N>.so (-DMK_ABLATE=N, see scan_kernel_impl.hpp) for the
# breakdown scripts; run in the repository root before gpurun (hipcc cross-compiles without a GPU)
# usage: tools/build_ablations.sh [N ...]      default: 1 7 8 16 32 64 256
set -e
cd "$(dirname "$0")/.."
for a in ${@:-1 7 8 16 32 64 256}; do
  python -m merkurio_amd.build --tag abl$a --flags "-DMK_ABLATE=$a" > /dev/null
done
ls -la merkurio_amd/lib/libmerkurio_hip_abl*.so
