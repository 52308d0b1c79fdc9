#!/bin/bash
# round 2, GPU session 20: full GPU suite (incl. graph capture, configuration fuzz).
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02t"
mkdir -p "$out"
cd "$root"
echo "== tests"; timeout -k 10 1000 python -m pytest tests -m gpu -q -rA -s > "$out/gpu_tests.log" 2>&1; rc=$?; tail -2 "$out/gpu_tests.log"
grep -E "^(FAILED|ERROR)" "$out/gpu_tests.log" | head -30
