#!/bin/bash
# the feature tiers a user runs, shipped library, through rollout() on two queues at 2^20 envs (C3 / C2 at their BASELINE sizes): two rounds
cd "$(dirname "$0")/.." || exit 1
out="gpurun_out/r04/tiers"; mkdir -p "$out"
run() { name="$1"; steps="$2"; shift 2
  a=$(timeout -k 10 300 python bench.py --steps "$steps" --warmup 100 --no-cpu-baseline --no-sub-records "$@" 2>/dev/null | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['roofline']['kernel_us_per_launch'])")
  b=$(timeout -k 10 300 python bench.py --steps "$steps" --warmup 100 --no-cpu-baseline --no-sub-records "$@" 2>/dev/null | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['roofline']['kernel_us_per_launch'])")
  printf "%-44s %s / %s us per step\n" "$name" "$a" "$b"; }
{
run "headline (C4 + episode returns)" 1500
run "default make() info (info_level=full)" 1000 --env-kw info_level=full
run "goal horizon 1, rl_reward" 1000 --env-kw cost=rl_reward --env-kw obs_goal_horizon=1
run "normalised actions, rl_reward" 1000 --env-kw normalized_rl_action_space=true --env-kw cost=rl_reward
run "white observation noise" 600 --env-kw "disturbances={observation: [{disturbance_func: white_noise, std: 0.01}]}"
run "white action noise" 600 --env-kw "disturbances={action: [{disturbance_func: white_noise, std: 0.001}]}"
run "constraints (input default + 6 state bounds)" 600 --env-kw "constraints=[{constraint_form: default_constraint, constrained_variable: input}, {constraint_form: bounded_constraint, constrained_variable: state, active_dims: [0, 2, 4], lower_bounds: [-3, -3, -0.1], upper_bounds: [3, 3, 2]}]"
run "randomised inertial properties" 600 --env-kw randomized_inertial_prop=true
run "precise mode (full)" 600 --env-kw compensated_state=true
run "precise mode (attitude)" 600 --env-kw compensated_state=attitude
run "explicit Euler" 1000 --env-kw physics=euler
run "C3 (2^18)" 1500 --workload c3 --envs-per-gpu 262144
run "C2 (2^16)" 1500 --workload c2 --envs-per-gpu 65536
} 2>&1 | tee "$out/feature_tiers.txt"
