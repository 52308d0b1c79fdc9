#!/bin/bash
# Build libscg_rollout.so (HIP kernels + C ABI) for gfx950, in-tree.
# -fno-slp-vectorize: the SLP vectoriser packs the RK4 stages into v_pk_fma_f32 register pairs,
# which costs 40+ VGPRs (147 vs 103) and a wave of occupancy per SIMD for no FLOP-rate gain.
# The translation units are compiled in parallel (scg_kernels.hip dominates: ~90 s), then linked.  A unit is recompiled
# when its source, any header of this directory, the C-ABI header or the flags changed (SCG_BUILD_FORCE=1: always).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../lib"
obj="$out/obj"
mkdir -p "$out" "$obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
flags=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@")
units=(scg_kernels scg_gae scg_linearize scg_normalize scg_expm scg_flow scg_capi)
stamp="$obj/flags.txt"
if [ ! -f "$stamp" ] || [ "$(cat "$stamp")" != "${flags[*]}" ]; then rm -f "$obj"/*.o; echo "${flags[*]}" > "$stamp"; fi
pids=()
for u in "${units[@]}"; do
    stale=0
    if [ "${SCG_BUILD_FORCE:-0}" = "1" ] || [ ! -f "$obj/$u.o" ]; then stale=1; else
        for dep in "$here/$u.hip" "$here"/*.h "$here/../../include/scg_rollout.h"; do
            if [ "$dep" -nt "$obj/$u.o" ]; then stale=1; fi
        done
    fi
    if [ "$stale" = "1" ]; then
        "$HIPCC" "${flags[@]}" -c "$here/$u.hip" -o "$obj/$u.o" &
        pids+=($!)
    fi
done
for p in ${pids[@]+"${pids[@]}"}; do wait "$p"; done
objs=()
for u in "${units[@]}"; do objs+=("$obj/$u.o"); done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$out/libscg_rollout.so" "${objs[@]}"
echo "built $out/libscg_rollout.so"
