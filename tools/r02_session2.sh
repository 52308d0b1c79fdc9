#!/bin/bash
# round 2, GPU session 2: full GPU suite (new: pins, expm, distributions), then a first bench of the new paths.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02b"
mkdir -p "$out"
cd "$root"
echo "== tests"; timeout -k 10 1000 python -m pytest tests -m gpu -q > "$out/gpu_tests.log" 2>&1; rc=$?; tail -5 "$out/gpu_tests.log"
grep -E "^(FAILED|ERROR)" "$out/gpu_tests.log" | head -30
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "test run ended abnormally (rc=$rc): stop"; exit $rc; fi
echo "== smoke"; python __graft_entry__.py smoke 2>&1 | tail -1
echo "== expm bench"; python tools/bench_expm.py > "$out/bench_expm.json" 2> "$out/bench_expm.err"; tail -3 "$out/bench_expm.json"
