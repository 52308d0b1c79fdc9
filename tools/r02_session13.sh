#!/bin/bash
# round 2, GPU session 13: fresh outputs by ping-pong state (scg_step_io.state_out): full suite + Python step API timings.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02m"
mkdir -p "$out"
cd "$root"
echo "== tests"; timeout -k 10 1000 python -m pytest tests -m gpu -q -rA -s > "$out/gpu_tests.log" 2>&1; rc=$?; tail -2 "$out/gpu_tests.log"
grep -E "^(FAILED|ERROR)" "$out/gpu_tests.log" | head -30
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "abnormal end rc=$rc"; exit $rc; fi
py() { name="$1"; shift; python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-sub-records --python-loop "$@" > "$out/bench_pyloop_$name.json" 2> "$out/bench_pyloop_$name.err" || tail -3 "$out/bench_pyloop_$name.err"; python -c "import json; j=json.load(open('$out/bench_pyloop_$name.json')); print('%-34s rollout %.2f us/step, python step() loop %.2f us/step' % ('$name', j['ms_per_step']*1e3, j['envs_per_gpu']/j['python_step_api_value']*1e6))"; }
echo "== python step API at 2^20 envs"
py full_fresh --env-kw info_level=full --env-kw copy_outputs=true
py full_views --env-kw info_level=full --env-kw copy_outputs=false
py minimal_views
py minimal_fresh --env-kw copy_outputs=true
py full_fresh_goal_horizon1 --env-kw info_level=full --env-kw copy_outputs=true --env-kw normalized_rl_action_space=true --env-kw cost=rl_reward --env-kw obs_goal_horizon=1
py full_views_goal_horizon1 --env-kw info_level=full --env-kw copy_outputs=false --env-kw normalized_rl_action_space=true --env-kw cost=rl_reward --env-kw obs_goal_horizon=1
echo "== headline"; python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records > "$out/bench_driver_cmd_nosub.json" 2>/dev/null; python -c "import json; j=json.load(open('$out/bench_driver_cmd_nosub.json')); print('driver args: %.2f us' % (j['ms_per_step']*1e3))"
python examples/ppo_rollout.py 262144 64 2>&1 | tail -2
