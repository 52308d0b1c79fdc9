#!/bin/bash
# Round-3 session on the final tree: the drift gates with their printed numbers, the new GPU tests, and the feature tiers against the
# exported round-2 tree (tools/r02_tree, see tools/tune.py --tree).   usage: tools/r03_final.sh <tag>
set -u
tag="${1:-final1}"
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r03/$tag"; mkdir -p "$out"; cd "$root"
python -m pytest tests/test_parity_gpu.py -q -m gpu -s -k "drift" 2>&1 | grep -E "drift|passed|failed" | tee "$out/drift_gates.txt"
python -m pytest tests/test_properties_gpu.py -q -m gpu -k "non_finite or graph or fresh" 2>&1 | tail -2
T="timeout -k 10 600 python tools/tune.py --rounds 2 --arms r02,r03 --tree r02=tools/r02_tree"
echo "== default make() tier (info_level=full), rollout"; $T --steps 1000 -- --env-kw info_level=full 2>&1 | tee "$out/tune_info.log" | grep -A4 "^----"
echo "== default make() tier, Python step loop (env-steps/s)"; $T --steps 1000 --field python_step_api_value -- --python-loop --env-kw info_level=full 2>&1 | tee "$out/tune_info_python.log" | grep -A4 "^----"
echo "== lean tier, Python step loop (env-steps/s)"; $T --steps 1000 --field python_step_api_value -- --python-loop 2>&1 | tee "$out/tune_lean_python.log" | grep -A4 "^----"
echo "== rl_reward + obs_goal_horizon=1"; $T --steps 1000 -- --env-kw cost=rl_reward --env-kw obs_goal_horizon=1 2>&1 | tee "$out/tune_goal_horizon.log" | grep -A4 "^----"
echo "== rl_reward + obs_goal_horizon=1, Python step loop, fresh outputs"; $T --steps 1000 --field python_step_api_value -- --python-loop --env-kw cost=rl_reward --env-kw obs_goal_horizon=1 --env-kw info_level=full 2>&1 | tee "$out/tune_goal_horizon_python.log" | grep -A4 "^----"
echo "== normalised actions"; $T --steps 1000 -- --env-kw normalized_rl_action_space=true --env-kw cost=rl_reward 2>&1 | tee "$out/tune_normalized.log" | grep -A4 "^----"
