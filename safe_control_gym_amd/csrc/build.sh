#!/bin/bash
# Build libscg_rollout.so (HIP kernels + C ABI) for gfx950, in-tree.
# -fno-slp-vectorize: the SLP vectoriser packs the RK4 stages into v_pk_fma_f32 register pairs,
# which costs 40+ VGPRs (147 vs 103) and a wave of occupancy per SIMD for no FLOP-rate gain.
# The translation units are compiled in parallel (scg_kernels.hip dominates: ~60 s), then linked.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../lib"
obj="$out/obj"
mkdir -p "$out" "$obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
flags=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@")
units=(scg_kernels scg_gae scg_linearize scg_normalize scg_expm scg_capi)
pids=()
for u in "${units[@]}"; do
    "$HIPCC" "${flags[@]}" -c "$here/$u.hip" -o "$obj/$u.o" &
    pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
objs=()
for u in "${units[@]}"; do objs+=("$obj/$u.o"); done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$out/libscg_rollout.so" "${objs[@]}"
echo "built $out/libscg_rollout.so"
