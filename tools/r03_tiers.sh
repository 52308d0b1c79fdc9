#!/bin/bash
# Round-3 session: the kernel-body arms of lib/variants on the VALU-bound configurations (fused rollout, default make() tier through
# the Python step loop, goal-horizon RL observation, observation noise).  usage: tools/r03_tiers.sh <tag>
set -u
tag="${1:-tiers1}"; shift || true
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r03/$tag"
mkdir -p "$out"
cd "$root"
T="timeout -k 10 600 python tools/tune.py --rounds ${ROUNDS:-2} ${ARMS:+--arms $ARMS} ${ABI:+--abi $ABI} ${TREE:+--tree $TREE}"
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  echo "== GPU suite (shipped library)"; timeout -k 10 900 python -m pytest tests -m gpu -q > "$out/gpu_tests.log" 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" "$out/gpu_tests.log" | tail -15
fi
echo "== headline, two queues"; $T --steps 1500 2>&1 | tee "$out/tune_headline.log" | grep -A20 "^----"
echo "== fused rollout"; $T --steps 600 --field fused_rollout.us_per_step -- --fused 2>&1 | tee "$out/tune_fused.log" | grep -A20 "^----"
echo "== default make() tier (info_level=full), rollout"; $T --steps 1000 -- --env-kw info_level=full 2>&1 | tee "$out/tune_info.log" | grep -A20 "^----"
echo "== default make() tier, Python step loop (env-steps/s)"; $T --steps 1000 --field python_step_api_value -- --python-loop --env-kw info_level=full 2>&1 | tee "$out/tune_info_python.log" | grep -A20 "^----"
echo "== rl_reward + obs_goal_horizon=1"; $T --steps 1000 -- --env-kw cost=rl_reward --env-kw obs_goal_horizon=1 2>&1 | tee "$out/tune_goal_horizon.log" | grep -A20 "^----"
echo "== white observation noise"; $T --steps 600 -- --env-kw "disturbances={observation: [{disturbance_func: white_noise, std: 0.01}]}" 2>&1 | tee "$out/tune_obs_noise.log" | grep -A20 "^----"
echo "== precise attitude"; $T --steps 600 -- --env-kw compensated_state=attitude 2>&1 | tee "$out/tune_attitude.log" | grep -A20 "^----"
