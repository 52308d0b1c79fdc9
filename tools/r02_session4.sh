#!/bin/bash
# round 2, GPU session 4: full suite after the FT_DIST tier and the attitude precise mode; driver command; disturbed-step decomposition.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02d"
mkdir -p "$out"
cd "$root"
echo "== tests"; timeout -k 10 1000 python -m pytest tests -m gpu -q > "$out/gpu_tests.log" 2>&1; rc=$?; tail -4 "$out/gpu_tests.log"
grep -E "^(FAILED|ERROR)" "$out/gpu_tests.log" | head -30
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "test run ended abnormally (rc=$rc): stop"; exit $rc; fi
echo "== bench, the driver's command"; python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_cmd.json" 2> "$out/bench_driver_cmd.err" || tail -5 "$out/bench_driver_cmd.err"
python - "$out/bench_driver_cmd.json" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print('main %.2f us (frac %.3f)' % (j['ms_per_step'] * 1e3, j['roofline']['frac']))
for k in ('c4_without_episode_stats', 'precise_mode', 'precise_mode_attitude', 'c3', 'c2'):
    if k in j: print('%-28s %.2f us  frac %.3f' % (k, j[k]['us_per_step'], j[k]['frac']))
PY
echo "== disturbed-step decomposition (2^20 envs, 500 steps)"
run() { name="$1"; shift; python bench.py --steps 500 --warmup 50 --no-cpu-baseline "$@" > "$out/decomp_$name.json" 2>/dev/null; python -c "import json; j=json.load(open('$out/decomp_$name.json')); print('%-42s %.2f us/step' % ('$name', j['ms_per_step']*1e3))"; }
run info_full --env-kw info_level=full
run extra_tier_no_noise --env-kw info_level=full --env-kw normalized_rl_action_space=true
run obs_white --env-kw info_level=full --env-kw 'disturbances={observation: [{disturbance_func: white_noise, std: 0.01}]}'
run action_white --env-kw info_level=full --env-kw 'disturbances={action: [{disturbance_func: white_noise, std: 0.002}]}'
run dynamics_uniform --env-kw info_level=full --env-kw 'disturbances={dynamics: [{disturbance_func: uniform, low: [-0.05, -0.05, -0.05], high: [0.05, 0.05, 0.05]}]}'
run all_three --env-kw info_level=full --env-kw 'disturbances={action: [{disturbance_func: white_noise, std: 0.002}], dynamics: [{disturbance_func: uniform, low: [-0.05, -0.05, -0.05], high: [0.05, 0.05, 0.05]}], observation: [{disturbance_func: white_noise, std: 0.01}]}'
run all_three_minimal_info --env-kw 'disturbances={action: [{disturbance_func: white_noise, std: 0.002}], dynamics: [{disturbance_func: uniform, low: [-0.05, -0.05, -0.05], high: [0.05, 0.05, 0.05]}], observation: [{disturbance_func: white_noise, std: 0.01}]}'
echo "== step-server probe (fixed checksum)"
for n in 16384 65536; do timeout -k 5 120 tools/ubench/step_server $n 2000 >> "$out/step_server_probe.jsonl" 2>> "$out/step_server_probe.err" || break; done; cat "$out/step_server_probe.jsonl"
