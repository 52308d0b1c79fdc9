#!/bin/bash
# round 2, GPU session 26: new full-size property tests.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02x"
mkdir -p "$out"
cd "$root"
timeout -k 10 900 python -m pytest tests/test_gae.py tests/test_properties_gpu.py -m gpu -q -rA -s -k "full_size" > "$out/new_tests.log" 2>&1; tail -4 "$out/new_tests.log"; grep -n "^E  " "$out/new_tests.log" | head -10
