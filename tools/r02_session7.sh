#!/bin/bash
# round 2, GPU session 7: fused-rollout reset variant A/B, the re-run of the fixed test, final test log.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02g"
mkdir -p "$out"
cd "$root"
echo "== fused rollout: pooled vs per-wave reset inside the step loop"
python tools/tune.py --rounds 2 --steps 1500 --arms base,fusedcoop --field fused_rollout.us_per_step -- --fused > "$out/tune_fused_c4.log" 2>&1; tail -3 "$out/tune_fused_c4.log"
python tools/tune.py --rounds 2 --steps 4000 --arms base,fusedcoop --field fused_rollout.us_per_step -- --fused --workload c3 --envs-per-gpu 262144 > "$out/tune_fused_c3.log" 2>&1; tail -3 "$out/tune_fused_c3.log"
python tools/tune.py --rounds 2 --steps 8000 --arms base,fusedcoop --field fused_rollout.us_per_step -- --fused --workload c2 --envs-per-gpu 65536 > "$out/tune_fused_c2.log" 2>&1; tail -3 "$out/tune_fused_c2.log"
echo "== tests"; timeout -k 10 1000 python -m pytest tests -m gpu -q -rA -s > "$out/gpu_tests.log" 2>&1; rc=$?; tail -2 "$out/gpu_tests.log"
grep -E "^(FAILED|ERROR)" "$out/gpu_tests.log" | head -30
