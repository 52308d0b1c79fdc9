#!/bin/bash
# Time composition of the step kernel by switching features off through make() kwargs (tuning only).
run() { # label, bench args...
  local label="$1"; shift
  python bench.py --steps 1500 --no-cpu-baseline --no-sub-records "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%-44s %.2f us  %.1f%%' % ('$label', j['roofline']['kernel_us_per_launch'], 100*j['roofline']['frac']))"
}
run "default"
run "no-oob-done (no resets)" --env-kw done_on_out_of_bound=false --env-kw episode_len_sec=600
run "euler (1 f_eval)" --env-kw physics=euler
run "euler + no resets" --env-kw physics=euler --env-kw done_on_out_of_bound=false --env-kw episode_len_sec=600
run "no motor map" --env-kw motor_map=false
run "stabilization (no goal gather)" --env-kw task=stabilization --env-kw 'task_info={stabilization_goal: [0,0,1], stabilization_goal_tolerance: 0.0}'
run "substeps=2" --env-kw integration_substeps=2
run "one launch queue" --queues 1
