#!/bin/bash
# round 2, GPU session 19: hipGraph replay vs plain launches at the BASELINE sizes.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02s"
mkdir -p "$out"
cd "$root"
timeout -k 10 600 python tools/graph_probe.py > "$out/graph_probe.txt" 2> "$out/graph_probe.err" || tail -5 "$out/graph_probe.err"
cat "$out/graph_probe.txt"
