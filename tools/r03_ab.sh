#!/bin/bash
# Round-3 A/B session on the GPU box: the GPU suite on the shipped library, then the kernel-body arms of lib/variants interleaved
# (tools/tune.py), then SQ counters (VALU instructions per env-step) of chosen arms.  usage: tools/r03_ab.sh <tag> [arms for pmc...]
set -u
tag="${1:-ab1}"; shift || true
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r03/$tag"
mkdir -p "$out"
cd "$root"
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  echo "== GPU suite (shipped library)"; timeout -k 10 900 python -m pytest tests -m gpu -q -x > "$out/gpu_tests.log" 2>&1; tail -3 "$out/gpu_tests.log"
fi
echo "== arms, two launch queues, 1500 steps"; timeout -k 10 900 python tools/tune.py --rounds ${ROUNDS:-3} --steps 1500 ${ARMS:+--arms $ARMS} 2>&1 | tee "$out/tune_two_queues.log" | tail -12
if [ "${ONE_QUEUE:-1}" = "1" ]; then
  echo "== arms, one launch queue"; timeout -k 10 900 python tools/tune.py --rounds 2 --steps 1500 ${ARMS:+--arms $ARMS} -- --queues 1 2>&1 | tee "$out/tune_one_queue.log" | tail -8
fi
for arm in "$@"; do
  echo "== SQ counters, arm $arm"
  SCG_ROLLOUT_LIB="$root/safe_control_gym_amd/lib/variants/$arm.so" PMC_GROUPS_ONLY="SQ" profiles/pmc.sh "$out/pmc_$arm" > "$out/pmc_summary_$arm.txt" 2>&1
  grep -E "SQ_INSTS_VALU|SQ_WAVES|SQ_INSTS_SALU|SQ_INSTS_LDS|SQ_WAIT_ANY|SQ_ACTIVE_INST_VALU|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES" "$out/pmc_summary_$arm.txt"
  rm -rf "$out/pmc_$arm"/pass*
done
