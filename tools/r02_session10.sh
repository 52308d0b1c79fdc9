#!/bin/bash
# round 2, GPU session 10: normalised actions in every tier, goal horizon in the FT_DIST tier (the usual RL configurations).
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02j"
mkdir -p "$out"
cd "$root"
echo "== tests"; timeout -k 10 1000 python -m pytest tests -m gpu -q -rA -s > "$out/gpu_tests.log" 2>&1; rc=$?; tail -2 "$out/gpu_tests.log"
grep -E "^(FAILED|ERROR)" "$out/gpu_tests.log" | head -30
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "abnormal end rc=$rc"; exit $rc; fi
run() { name="$1"; shift; python bench.py --steps 500 --warmup 50 --no-cpu-baseline "$@" > "$out/rl_$name.json" 2>/dev/null; python -c "import json; j=json.load(open('$out/rl_$name.json')); print('%-46s %.2f us/step' % ('$name', j['ms_per_step']*1e3))"; }
echo "== the headline (a run-time branch was added to the lean tiers)"; python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records > "$out/bench_driver_cmd_nosub.json" 2>/dev/null; python -c "import json; j=json.load(open('$out/bench_driver_cmd_nosub.json')); print('driver args: %.2f us' % (j['ms_per_step']*1e3))"
run default_3000 --steps 3000 --no-sub-records
echo "== RL configurations at 2^20 envs"
run normalized_actions_info_full --env-kw info_level=full --env-kw copy_outputs=false --env-kw normalized_rl_action_space=true
run rl_track_goal_horizon1_normalized --env-kw info_level=full --env-kw copy_outputs=false --env-kw normalized_rl_action_space=true --env-kw cost=rl_reward --env-kw obs_goal_horizon=1
run rl_track_goal_horizon1_normalized_fresh --env-kw info_level=full --env-kw normalized_rl_action_space=true --env-kw cost=rl_reward --env-kw obs_goal_horizon=1
python examples/ppo_rollout.py 262144 64 2>&1 | tail -2
