#!/bin/bash
# Build libscg_rollout.so (HIP kernels + C ABI) for gfx950, in-tree.
# -fno-slp-vectorize: the SLP vectoriser packs the RK4 stages into v_pk_fma_f32 register pairs,
# which costs 40+ VGPRs (147 vs 103) and a wave of occupancy per SIMD for no FLOP-rate gain.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../lib"
mkdir -p "$out"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize \
    -o "$out/libscg_rollout.so" "$here/scg_kernels.hip" "$here/scg_gae.hip" "$here/scg_linearize.hip" "$here/scg_normalize.hip" "$here/scg_capi.hip" "$@"
echo "built $out/libscg_rollout.so"
