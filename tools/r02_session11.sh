#!/bin/bash
# round 2, GPU session 11: memory-floor probe with an LDS-transposed float4 variant, the new full-size linear-model test,
# the goal-horizon RL configurations.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02k"
mkdir -p "$out"
cd "$root"
echo "== new test"; timeout -k 10 600 python -m pytest tests/test_linearize.py -m gpu -q -k "full_size" > "$out/test_new.log" 2>&1; tail -3 "$out/test_new.log"
echo "== streams probe"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/streams tools/ubench/streams.hip > /dev/null 2>&1 && timeout -k 10 120 /tmp/streams > "$out/streams_probe.txt" 2>&1; cat "$out/streams_probe.txt"
run() { name="$1"; shift; python bench.py --steps 500 --warmup 50 --no-cpu-baseline "$@" > "$out/rl_$name.json" 2> "$out/rl_$name.err" || tail -3 "$out/rl_$name.err"; python -c "import json; j=json.load(open('$out/rl_$name.json')); print('%-46s %.2f us/step' % ('$name', j['ms_per_step']*1e3))"; }
echo "== RL configurations at 2^20 envs"
run rl_track_goal_horizon1_normalized --env-kw info_level=full --env-kw copy_outputs=false --env-kw normalized_rl_action_space=true --env-kw cost=rl_reward --env-kw obs_goal_horizon=1
run rl_track_goal_horizon1_normalized_fresh --env-kw info_level=full --env-kw normalized_rl_action_space=true --env-kw cost=rl_reward --env-kw obs_goal_horizon=1
