#!/bin/bash
# round 2, GPU session 31: attribute-surface test + full suite.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02ab"
mkdir -p "$out"
cd "$root"
timeout -k 10 1000 python -m pytest tests -m gpu -q -rA -s > "$out/gpu_tests.log" 2>&1; tail -2 "$out/gpu_tests.log"
grep -E "^(FAILED|ERROR)" "$out/gpu_tests.log" | head -20; grep -n "^E  " "$out/gpu_tests.log" | head -10
