#!/bin/bash
# round 2, GPU session 36: the configuration fuzz with 800 seeds (a one-off hunt; the suite runs 96).
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02af"
mkdir -p "$out"
cd "$root"
SCG_FUZZ_SEEDS=800 timeout -k 10 1100 python -m pytest tests/test_config_fuzz_gpu.py -m gpu -q -x --no-header -p no:cacheprovider > "$out/fuzz800.log" 2>&1; tail -5 "$out/fuzz800.log"; grep -n "^E  " "$out/fuzz800.log" | head -10
