#!/bin/bash
# round 2, GPU session 40: rocprofv3 kernel stats of the next-row kernels (GAE, Jacobians, normalisers, matrix exponential).
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02ah"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for t in gae normalize expm; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_$t" -- python3 "$root/tools/bench_$t.py" > "$out/prof_$t.log" 2>&1
  find "$out/prof_$t" -name "*kernel_stats.csv" -exec cp {} "$out/${t}_kernel_stats.csv" \;
  head -4 "$out/${t}_kernel_stats.csv" | cut -c1-200
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_linearize" -- python3 "$root/tools/bench_linearize.py" 3 > "$out/prof_linearize.log" 2>&1
find "$out/prof_linearize" -name "*kernel_stats.csv" -exec cp {} "$out/linearize_kernel_stats.csv" \;
head -3 "$out/linearize_kernel_stats.csv" | cut -c1-200
rm -rf "$out"/prof_*/
