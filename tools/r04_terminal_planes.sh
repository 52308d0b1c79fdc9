#!/bin/bash
# terminal_layout='planes' against 'records': the default-info step (rollout on two queues and the lone env.step() loop), two rounds
cd "$(dirname "$0")/.." || exit 1
out="gpurun_out/r04/terminal_planes"; mkdir -p "$out"
run() { name="$1"; shift
  a=$(timeout -k 10 300 python bench.py --steps 600 --warmup 100 --no-cpu-baseline --no-sub-records --python-loop "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f / loop %.2f' % (r['roofline']['kernel_us_per_launch'], n_ / r['python_step_api_value'] * 1e6 if (n_ := r['config'].get('envs_per_gpu', 1048576)) and r.get('python_step_api_value') else float('nan')))")
  printf "%-50s %s us per step\n" "$name" "$a"; }
{
for r in 1 2; do
run "default info, records" --env-kw info_level=full
run "default info, planes" --env-kw info_level=full --env-kw terminal_layout=planes
run "goal horizon + info, records" --env-kw info_level=full --env-kw cost=rl_reward --env-kw obs_goal_horizon=1
run "goal horizon + info, planes" --env-kw info_level=full --env-kw cost=rl_reward --env-kw obs_goal_horizon=1 --env-kw terminal_layout=planes
done
} 2>&1 | tee "$out/table.txt"
