#!/bin/bash
# the per-env-inertia tier (FT_INERT): parity / fuzz tests that randomise the inertial properties, then its step time (lean and default info)
cd "$(dirname "$0")/.." || exit 1
out="gpurun_out/r04/inert"; mkdir -p "$out"
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_config_fuzz_gpu.py tests/test_properties_gpu.py tests/test_disturbances.py tests/test_pipelines_gpu.py tests/test_gae.py -m gpu -q -x 2>&1 | tail -3 || exit 1
run() { name="$1"; shift
  a=$(timeout -k 10 300 python bench.py --steps 600 --warmup 100 --no-cpu-baseline --no-sub-records "$@" 2>/dev/null | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['roofline']['kernel_us_per_launch'])")
  b=$(timeout -k 10 300 python bench.py --steps 600 --warmup 100 --no-cpu-baseline --no-sub-records "$@" 2>/dev/null | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['roofline']['kernel_us_per_launch'])")
  printf "%-60s %s / %s us per step\n" "$name" "$a" "$b"; }
{
run "randomised inertial properties (lean info)" --env-kw randomized_inertial_prop=true
run "randomised inertial properties, default make() info" --env-kw randomized_inertial_prop=true --env-kw info_level=full
run "randomised inertial properties + goal horizon" --env-kw randomized_inertial_prop=true --env-kw cost=rl_reward --env-kw obs_goal_horizon=1
run "randomised inertial properties + white action noise" --env-kw randomized_inertial_prop=true --env-kw "disturbances={action: [{disturbance_func: white_noise, std: 0.001}]}"
run "headline" 
} 2>&1 | tee "$out/inertial_tier.txt"
