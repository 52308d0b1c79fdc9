#!/bin/bash
# round 2, GPU session 18: configuration fuzz against the oracle.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02r"
mkdir -p "$out"
cd "$root"
timeout -k 10 900 python -m pytest tests/test_config_fuzz_gpu.py -m gpu -q -rA -s > "$out/fuzz.log" 2>&1; tail -3 "$out/fuzz.log"
grep -E "^(FAILED|ERROR)" "$out/fuzz.log" | cut -c1-400 | head -40
