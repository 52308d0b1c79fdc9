#!/bin/bash
# Time composition of the step kernel by switching features off through make() kwargs (tuning only).
run() { # label, env, args...
  local label="$1"; shift
  python bench.py --steps 1500 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%-44s %.2f us  %.1f%%' % ('$label', j['roofline']['kernel_us_per_launch'], 100*j['roofline']['frac']))"
}
for pk in 1 0; do
  export SCG_PACKED=$pk
  run "packed=$pk default"
  run "packed=$pk no-oob-done (no resets)" --env-kw done_on_out_of_bound=false --env-kw episode_len_sec=600
  run "packed=$pk euler (1 f_eval)" --env-kw physics=euler
  run "packed=$pk euler + no resets" --env-kw physics=euler --env-kw done_on_out_of_bound=false --env-kw episode_len_sec=600
  run "packed=$pk no motor map" --env-kw motor_map=false
  run "packed=$pk stabilization (no goal gather)" --env-kw task=stabilization --env-kw 'task_info={stabilization_goal: [0,0,1], stabilization_goal_tolerance: 0.0}'
  run "packed=$pk substeps=2" --env-kw integration_substeps=2
done
