#!/bin/bash
# CPU sanitizer build of the C-ABI validation layer: scg_capi.hip compiled as plain host C++ against the stand-in HIP
# runtime of this directory (GPU sanitizers are not available on this pool; sanitizers belong on the CPU build).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
root="$here/../.."
out="$here/_build"
mkdir -p "$out"
g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=all -Wall -Wextra \
    -I "$here" -x c++ "$root/safe_control_gym_amd/csrc/scg_capi.hip" "$here/hip_host_shim.cc" "$here/capi_sanitize_main.cc" \
    -o "$out/capi_sanitize"
echo "built $out/capi_sanitize"
