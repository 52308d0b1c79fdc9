#!/bin/bash
# round 2, GPU session 16: smoke(), full suite, host overhead after exact-size fresh outputs.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02p"
mkdir -p "$out"
cd "$root"
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
echo "== tests"; timeout -k 10 1000 python -m pytest tests -m gpu -q -rA -s > "$out/gpu_tests.log" 2>&1; rc=$?; tail -2 "$out/gpu_tests.log"
grep -E "^(FAILED|ERROR)" "$out/gpu_tests.log" | head -30
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "abnormal end rc=$rc"; exit $rc; fi
python tools/step_host_overhead.py 4096 3000 2>&1 | grep "us per step" > "$out/step_host_overhead_4096.txt"; cat "$out/step_host_overhead_4096.txt"
python tools/step_host_overhead.py 262144 2000 2>&1 | grep "us per step" > "$out/step_host_overhead_262144.txt"; cat "$out/step_host_overhead_262144.txt"
