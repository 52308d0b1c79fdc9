#!/bin/bash
# round 2, GPU session 33: examples.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02ac"
mkdir -p "$out"
cd "$root"
timeout -k 10 600 python -m pytest tests/test_examples_gpu.py -m gpu -q -rA -s > "$out/examples.log" 2>&1; tail -3 "$out/examples.log"; grep -n "^E  " "$out/examples.log" | head
python examples/linearize_along_trajectory.py 1024 2>&1 | tail -3
