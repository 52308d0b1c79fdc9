#!/bin/bash
# Build A/B variants of the HIP library into safe_control_gym_amd/lib/variants/ (tuning only).  Only scg_kernels.hip is
# recompiled per variant; the other translation units come from lib/obj (run csrc/build.sh first).
# usage: tools/build_variants.sh            -> the default set below
#        tools/build_variants.sh name "flags" [name "flags" ...]
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
src="$root/safe_control_gym_amd/csrc"
obj="$root/safe_control_gym_amd/lib/obj"
out="$root/safe_control_gym_amd/lib/variants"
mkdir -p "$out"
rm -f "$out"/*.o        # arms already in lib/variants stay (e.g. a copy of an earlier round's library)
build() { # name, extra flags...
  local name="$1"; shift
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c "$src/scg_kernels.hip" -o "$out/$name.o" &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/$name.so" "$out/$name.o" "$obj/scg_gae.o" "$obj/scg_linearize.o" "$obj/scg_normalize.o" "$obj/scg_expm.o" "$obj/scg_flow.o" "$obj/scg_capi.o" &&
    rm -f "$out/$name.o" ) &
}
if [ "$#" -gt 0 ]; then
  while [ "$#" -gt 1 ]; do n="$1"; f="$2"; shift 2; build "$n" $f; done
else
  build r1kernel -fno-slp-vectorize -DSCG_WG_RESET=0 -DSCG_OOB_MASK=0
  build wg -fno-slp-vectorize -DSCG_WG_RESET=1 -DSCG_OOB_MASK=0
  build wg_oob -fno-slp-vectorize
  build wg_oob_sg80 -fno-slp-vectorize -DSCG_NUM_SGPR=80
  build wg_oob_slp
  build wg_oob_sg80_slp -DSCG_NUM_SGPR=80
fi
wait
ls -la "$out"
