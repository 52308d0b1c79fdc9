#!/bin/bash
# round 2, GPU session 9: software-pipelined expm kernel (tests + bench), launcher test.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02i"
mkdir -p "$out"
cd "$root"
echo "== tests"; timeout -k 10 1000 python -m pytest tests -m gpu -q -rA -s > "$out/gpu_tests.log" 2>&1; rc=$?; tail -2 "$out/gpu_tests.log"
grep -E "^(FAILED|ERROR)" "$out/gpu_tests.log" | head -30
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "abnormal end rc=$rc"; exit $rc; fi
echo "== expm bench"; python tools/bench_expm.py > "$out/bench_expm.json" 2>/dev/null; cut -c1-330 "$out/bench_expm.json"
