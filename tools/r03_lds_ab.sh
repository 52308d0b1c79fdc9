#!/bin/bash
# A/B on one box, interleaved: X_GOAL gathered from global memory (ref_in_lds=0) vs staged in LDS (1), goal-horizon RL configuration.
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$root"; mkdir -p gpurun_out/r03
for r in 1 2 3; do for v in 0 1; do
  python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-sub-records --env-kw cost=rl_reward --env-kw obs_goal_horizon=1 --env-kw ref_in_lds=$v 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('round $r ref_in_lds=$v  %.2f us/step' % (j['ms_per_step']*1e3))"
done; done | tee gpurun_out/r03/goal_horizon_ref_in_lds_ab.txt
for r in 1 2; do for v in 0 1; do
  python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-sub-records --env-kw ref_in_lds=$v 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('headline round $r ref_in_lds=$v  %.2f us/step' % (j['ms_per_step']*1e3))"
done; done | tee -a gpurun_out/r03/goal_horizon_ref_in_lds_ab.txt
