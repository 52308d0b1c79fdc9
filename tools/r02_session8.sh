#!/bin/bash
# round 2, GPU session 8: expm with LDS-staged output, full suite after the fused-rollout reset change.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02h"
mkdir -p "$out"
cd "$root"
echo "== tests"; timeout -k 10 1000 python -m pytest tests -m gpu -q -rA -s > "$out/gpu_tests.log" 2>&1; rc=$?; tail -2 "$out/gpu_tests.log"
grep -E "^(FAILED|ERROR)" "$out/gpu_tests.log" | head -30
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "abnormal end rc=$rc"; exit $rc; fi
echo "== expm bench"; python tools/bench_expm.py > "$out/bench_expm.json" 2>/dev/null; cut -c1-330 "$out/bench_expm.json"
echo "== fused"; python bench.py --no-cpu-baseline --no-sub-records --fused > "$out/bench_fused_c4.json" 2>/dev/null; python -c "import json; j=json.load(open('$out/bench_fused_c4.json')); print('c4 %.2f us, fused %.2f us/step' % (j['ms_per_step']*1e3, j['fused_rollout']['us_per_step']))"
