#!/bin/bash
# Build A/B variants of the HIP library into safe_control_gym_amd/lib/variants/ (tuning only).
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
src="$root/safe_control_gym_amd/csrc"
out="$root/safe_control_gym_amd/lib/variants"
mkdir -p "$out"
build() { # name, extra flags...
  local name="$1"; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o "$out/$name.so" "$src/scg_kernels.hip" "$src/scg_gae.hip" "$src/scg_linearize.hip" "$src/scg_normalize.hip" "$src/scg_capi.hip" &
}
build base -fno-slp-vectorize
build nt2 -fno-slp-vectorize -DSCG_NT=2
build b128 -fno-slp-vectorize -DSCG_BLOCK=128
build w5 -fno-slp-vectorize -DSCG_MIN_WAVES=5
build lin_ep8 -fno-slp-vectorize -DSCG_LIN_EP3=8
build gae1 -fno-slp-vectorize -DSCG_GAE_DEPTH=1
wait
ls -la "$out"
