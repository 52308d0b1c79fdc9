#!/bin/bash
# Regression sweep of round 3: every tier / workload of the shipped library against the exported round-2 tree, interleaved on one box
# (the disturbance tier slipped unnoticed once during this round: profiles/r03/dist_table_in_lds.txt).  usage: tools/r03_regress.sh [tag]
set -u
tag="${1:-regress}"
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r03/$tag"; mkdir -p "$out"; cd "$root"
T="timeout -k 10 400 python tools/tune.py --rounds 2 --arms r02,r03 --tree r02=tools/r02_tree"
run() { name="$1"; shift; echo "== $name"; $T "$@" 2>&1 | tee "$out/tune_$name.log" | grep -A3 "^----" | tail -2; }
run headline --steps 1500
run fused --steps 600 --field fused_rollout.us_per_step -- --fused
run info --steps 1000 -- --env-kw info_level=full
run goal_horizon --steps 1000 -- --env-kw cost=rl_reward --env-kw obs_goal_horizon=1
run normalized --steps 1000 -- --env-kw normalized_rl_action_space=true --env-kw cost=rl_reward
run obs_noise --steps 600 -- --env-kw "disturbances={observation: [{disturbance_func: white_noise, std: 0.01}]}"
run act_noise --steps 600 -- --env-kw "disturbances={action: [{disturbance_func: white_noise, std: 0.001}]}"
run constraints --steps 600 -- --env-kw "constraints=[{constraint_form: default_constraint, constrained_variable: input}, {constraint_form: bounded_constraint, constrained_variable: state, active_dims: [0, 2, 4], upper_bounds: [3, 3, 2], lower_bounds: [-3, -3, -0.1]}]"
run inertial --steps 600 -- --env-kw randomized_inertial_prop=true
run precise --steps 600 -- --env-kw compensated_state=true
run c3 --steps 1500 -- --workload c3 --envs-per-gpu 262144
run c2 --steps 1500 -- --workload c2 --envs-per-gpu 65536
run euler --steps 1000 -- --env-kw physics=euler
run stabilization --steps 1000 -- --env-kw task=stabilization --env-kw "task_info={stabilization_goal: [0, 0, 1], stabilization_goal_tolerance: 0.05}"
