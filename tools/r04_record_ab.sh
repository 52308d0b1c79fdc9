#!/bin/bash
# the goal-horizon record's store width (SCG_RECORD_VEC = 1 / 2 / 4 floats per store), interleaved on one box
cd "$(dirname "$0")/.." || exit 1
root=$(pwd); out="$root/gpurun_out/r04/record_ab"; mkdir -p "$out"
for r in 1 2 3; do
  for arm in rec1 rec2 rec4; do
    export SCG_ROLLOUT_LIB="$root/safe_control_gym_amd/lib/variants/$arm.so"
    a=$(timeout -k 10 300 python bench.py --steps 600 --warmup 100 --no-cpu-baseline --no-sub-records --env-kw info_level=full --env-kw cost=rl_reward --env-kw obs_goal_horizon=1 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % r['roofline']['kernel_us_per_launch'])")
    echo "round $r $arm: $a us per step (goal horizon + default info, 2^20 envs, i.i.d. random actions)"
  done
done 2>&1 | tee "$out/table.txt"
