#!/bin/bash
# round 2, GPU session 12: goal-horizon observation sharing the state planes (obs == state), full suite + timings.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02l"
mkdir -p "$out"
cd "$root"
echo "== tests"; timeout -k 10 1000 python -m pytest tests -m gpu -q -rA -s > "$out/gpu_tests.log" 2>&1; rc=$?; tail -2 "$out/gpu_tests.log"
grep -E "^(FAILED|ERROR)" "$out/gpu_tests.log" | head -30
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "abnormal end rc=$rc"; exit $rc; fi
run() { name="$1"; shift; python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-sub-records "$@" > "$out/rl_$name.json" 2> "$out/rl_$name.err" || tail -3 "$out/rl_$name.err"; python -c "import json; j=json.load(open('$out/rl_$name.json')); print('%-46s %.2f us/step' % ('$name', j['ms_per_step']*1e3))"; }
echo "== headline"; run default_3000 --steps 3000
echo "== RL configurations at 2^20 envs"
run rl_track_goal_horizon1_normalized_views --env-kw info_level=full --env-kw copy_outputs=false --env-kw normalized_rl_action_space=true --env-kw cost=rl_reward --env-kw obs_goal_horizon=1
run rl_track_goal_horizon1_normalized_views_minimal --env-kw copy_outputs=false --env-kw normalized_rl_action_space=true --env-kw cost=rl_reward --env-kw obs_goal_horizon=1
run rl_track_goal_horizon1_normalized_fresh --env-kw info_level=full --env-kw normalized_rl_action_space=true --env-kw cost=rl_reward --env-kw obs_goal_horizon=1
run rl_track_no_horizon_normalized_views --env-kw info_level=full --env-kw copy_outputs=false --env-kw normalized_rl_action_space=true --env-kw cost=rl_reward
