#!/bin/bash
# round 2, GPU session 3: re-run the tests fixed after session 2, the persistent step-server probe, the Python step API.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02c"
mkdir -p "$out"
cd "$root"
echo "== tests (subset)"; timeout -k 10 600 python -m pytest tests/test_linearize.py tests/test_randomization.py tests/test_reference_pins.py -m gpu -q > "$out/gpu_tests_subset.log" 2>&1; rc=$?; tail -3 "$out/gpu_tests_subset.log"
grep -E "^(FAILED|ERROR)" "$out/gpu_tests_subset.log" | head
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "abnormal end rc=$rc"; exit $rc; fi
echo "== step-server probe"
for n in 16384 65536 262144; do timeout -k 5 120 tools/ubench/step_server $n 2000 >> "$out/step_server_probe.jsonl" 2>> "$out/step_server_probe.err" || { echo "probe failed at n=$n"; tail -3 "$out/step_server_probe.err"; break; }; done
cat "$out/step_server_probe.jsonl"
echo "== python step API, default make() (full info, fresh outputs) vs views"
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --python-loop --env-kw info_level=full --env-kw copy_outputs=true > "$out/bench_pyloop_full_copy.json" 2>/dev/null; python -c "import json; j=json.load(open('$out/bench_pyloop_full_copy.json')); print('full+copy: rollout %.2f us/step, python loop %.3e env-steps/s = %.2f us/step' % (j['ms_per_step']*1e3, j['python_step_api_value'], 1048576/j['python_step_api_value']*1e6))"
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --python-loop --env-kw info_level=full --env-kw copy_outputs=false > "$out/bench_pyloop_full_views.json" 2>/dev/null; python -c "import json; j=json.load(open('$out/bench_pyloop_full_views.json')); print('full+views: rollout %.2f us/step, python loop %.3e env-steps/s = %.2f us/step' % (j['ms_per_step']*1e3, j['python_step_api_value'], 1048576/j['python_step_api_value']*1e6))"
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --python-loop --fused --no-sub-records > "$out/bench_pyloop_minimal.json" 2>/dev/null; python -c "import json; j=json.load(open('$out/bench_pyloop_minimal.json')); print('minimal: rollout %.2f us/step, python loop %.2f us/step, fused %.2f us/step' % (j['ms_per_step']*1e3, 1048576/j['python_step_api_value']*1e6, j['fused_rollout']['us_per_step']))"
