#!/bin/bash
# round 2, GPU session 5: precise-mode occupancy variants, the fixed precise-mode test, then the round's measurement set.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02e"
mkdir -p "$out"
cd "$root"
echo "== precise-mode tests"; timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "precise" > "$out/gpu_tests_precise.log" 2>&1; rc=$?; tail -3 "$out/gpu_tests_precise.log"; grep -E "^(FAILED|ERROR)|open-loop replay" "$out/gpu_tests_precise.log" | head
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "abnormal end rc=$rc"; exit $rc; fi
echo "== precise-mode occupancy variants: attitude mode"; python tools/tune.py --rounds 2 --steps 600 --arms base,cw5,cw6 -- --no-episode-stats --env-kw compensated_state=attitude > "$out/tune_comp_attitude.log" 2>&1; tail -4 "$out/tune_comp_attitude.log"
echo "== precise-mode occupancy variants: full mode"; python tools/tune.py --rounds 2 --steps 600 --arms base,cw5,cw6 -- --no-episode-stats --env-kw compensated_state=true > "$out/tune_comp_full.log" 2>&1; tail -4 "$out/tune_comp_full.log"
echo "== step-server probe"; for n in 16384 65536; do timeout -k 5 120 tools/ubench/step_server $n 2000 >> "$out/step_server_probe.jsonl" 2>> "$out/step_server_probe.err" || break; done; cat "$out/step_server_probe.jsonl"
