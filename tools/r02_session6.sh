#!/bin/bash
# round 2, GPU session 6: the full suite on the final code (log kept for profiles/r02), drift table, Python step API.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02f"
mkdir -p "$out"
cd "$root"
echo "== tests"; timeout -k 10 1000 python -m pytest tests -m gpu -q -rA -s > "$out/gpu_tests.log" 2>&1; rc=$?; tail -3 "$out/gpu_tests.log"
grep -E "^(FAILED|ERROR)" "$out/gpu_tests.log" | head -30
grep -h "open-loop replay\|closed-loop 1000" "$out/gpu_tests.log" | head
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "test run ended abnormally (rc=$rc): stop"; exit $rc; fi
echo "== smoke"; python __graft_entry__.py smoke 2>&1 | tail -1
echo "== python step API"
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --python-loop --env-kw info_level=full --env-kw copy_outputs=true > "$out/bench_python_api_full_copy.json" 2>/dev/null; python -c "import json; j=json.load(open('$out/bench_python_api_full_copy.json')); print('full + fresh outputs: rollout %.2f us/step, python loop %.2f us/step' % (j['ms_per_step']*1e3, 1048576/j['python_step_api_value']*1e6))"
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --python-loop --env-kw info_level=full --env-kw copy_outputs=false > "$out/bench_python_api_full_views.json" 2>/dev/null; python -c "import json; j=json.load(open('$out/bench_python_api_full_views.json')); print('full + views: rollout %.2f us/step, python loop %.2f us/step' % (j['ms_per_step']*1e3, 1048576/j['python_step_api_value']*1e6))"
python examples/ppo_rollout.py 262144 64 2>&1 | tail -2
