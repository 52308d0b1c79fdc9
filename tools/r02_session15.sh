#!/bin/bash
# round 2, GPU session 15: host cost of the Python step() call at small batches.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02o"
mkdir -p "$out"
cd "$root"
python tools/step_host_overhead.py 4096 3000 > "$out/step_host_overhead_4096.txt" 2>&1; cat "$out/step_host_overhead_4096.txt" | cut -c1-180
python tools/step_host_overhead.py 262144 2000 2>&1 | grep "us per step" > "$out/step_host_overhead_262144.txt"; cat "$out/step_host_overhead_262144.txt"
for i in 1 2 3; do python examples/ppo_rollout.py 262144 64 2>&1 | head -1; done
