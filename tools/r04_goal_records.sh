#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
timeout -k 10 900 python -m pytest tests/test_pipelines_gpu.py tests/test_parity_gpu.py tests/test_config_fuzz_gpu.py tests/test_reference_pins.py -m gpu -q -x 2>&1 | tail -3 || exit 1
run() { name="$1"; shift
  a=$(timeout -k 10 300 python bench.py --steps 600 --warmup 100 --no-cpu-baseline --no-sub-records "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % r['roofline']['kernel_us_per_launch'])")
  printf "%-50s %s us per step\n" "$name" "$a"; }
for r in 1 2; do
run "goal horizon + default info (records)" --env-kw info_level=full --env-kw cost=rl_reward --env-kw obs_goal_horizon=1
run "default info (records)" --env-kw info_level=full
done
