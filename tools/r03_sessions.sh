#!/bin/bash
# The GPU-box sessions of round 3 that produced the files under profiles/r03/ (one sub-command = one gpurun call = one box).
#   tools/r03_sessions.sh ab <tag> [arms to PMC...]   kernel-body arms interleaved (tools/tune.py) + SQ counters per arm
#   tools/r03_sessions.sh tiers <tag>                 the arms on the VALU-bound configurations (env: ARMS, TREE, ABI, ROUNDS, SKIP_TESTS)
#   tools/r03_sessions.sh misc <tag>                  region cost vs steps, C3 queues, N sweep with graph replay on / off
#   tools/r03_sessions.sh final <tag>                 drift gates with their numbers + the feature tiers against the round-2 tree
#   tools/r03_sessions.sh lds_ab                      X_GOAL gathered vs staged in LDS, interleaved
#   tools/r03_sessions.sh close                       full GPU suite, 400-seed configuration fuzz, the driver's command, smoke()
# Arms are library builds under safe_control_gym_amd/lib/variants/ (tools/build_variants.sh); the arm "r02" is the exported round-2 tree
# (git archive 88220ee safe_control_gym_amd bench.py profiles/hbm_traffic.json into tools/r02_tree/ + its library), see tools/tune.py --tree.
cmd="${1:-}"; shift || true
case "$cmd" in
ab)
(
# Round-3 A/B session on the GPU box: the GPU suite on the shipped library, then the kernel-body arms of lib/variants interleaved
# (tools/tune.py), then SQ counters (VALU instructions per env-step) of chosen arms.  usage: tools/r03_ab.sh <tag> [arms for pmc...]
set -u
tag="${1:-ab1}"; shift || true
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r03/$tag"
mkdir -p "$out"
cd "$root"
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  echo "== GPU suite (shipped library)"; timeout -k 10 900 python -m pytest tests -m gpu -q -x > "$out/gpu_tests.log" 2>&1; tail -3 "$out/gpu_tests.log"
fi
echo "== arms, two launch queues, 1500 steps"; timeout -k 10 900 python tools/tune.py --rounds ${ROUNDS:-3} --steps 1500 ${ARMS:+--arms $ARMS} 2>&1 | tee "$out/tune_two_queues.log" | tail -12
if [ "${ONE_QUEUE:-1}" = "1" ]; then
  echo "== arms, one launch queue"; timeout -k 10 900 python tools/tune.py --rounds 2 --steps 1500 ${ARMS:+--arms $ARMS} -- --queues 1 2>&1 | tee "$out/tune_one_queue.log" | tail -8
fi
for arm in "$@"; do
  echo "== SQ counters, arm $arm"
  SCG_ROLLOUT_LIB="$root/safe_control_gym_amd/lib/variants/$arm.so" PMC_GROUPS_ONLY="SQ" profiles/pmc.sh "$out/pmc_$arm" > "$out/pmc_summary_$arm.txt" 2>&1
  grep -E "SQ_INSTS_VALU|SQ_WAVES|SQ_INSTS_SALU|SQ_INSTS_LDS|SQ_WAIT_ANY|SQ_ACTIVE_INST_VALU|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES" "$out/pmc_summary_$arm.txt"
  rm -rf "$out/pmc_$arm"/pass*
done
)
;;
tiers)
(
# Round-3 session: the kernel-body arms of lib/variants on the VALU-bound configurations (fused rollout, default make() tier through
# the Python step loop, goal-horizon RL observation, observation noise).  usage: tools/r03_tiers.sh <tag>
set -u
tag="${1:-tiers1}"; shift || true
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r03/$tag"
mkdir -p "$out"
cd "$root"
T="timeout -k 10 600 python tools/tune.py --rounds ${ROUNDS:-2} ${ARMS:+--arms $ARMS} ${ABI:+--abi $ABI} ${TREE:+--tree $TREE}"
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  echo "== GPU suite (shipped library)"; timeout -k 10 900 python -m pytest tests -m gpu -q > "$out/gpu_tests.log" 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" "$out/gpu_tests.log" | tail -15
fi
echo "== headline, two queues"; $T --steps 1500 2>&1 | tee "$out/tune_headline.log" | grep -A20 "^----"
echo "== fused rollout"; $T --steps 600 --field fused_rollout.us_per_step -- --fused 2>&1 | tee "$out/tune_fused.log" | grep -A20 "^----"
echo "== default make() tier (info_level=full), rollout"; $T --steps 1000 -- --env-kw info_level=full 2>&1 | tee "$out/tune_info.log" | grep -A20 "^----"
echo "== default make() tier, Python step loop (env-steps/s)"; $T --steps 1000 --field python_step_api_value -- --python-loop --env-kw info_level=full 2>&1 | tee "$out/tune_info_python.log" | grep -A20 "^----"
echo "== rl_reward + obs_goal_horizon=1"; $T --steps 1000 -- --env-kw cost=rl_reward --env-kw obs_goal_horizon=1 2>&1 | tee "$out/tune_goal_horizon.log" | grep -A20 "^----"
echo "== white observation noise"; $T --steps 600 -- --env-kw "disturbances={observation: [{disturbance_func: white_noise, std: 0.01}]}" 2>&1 | tee "$out/tune_obs_noise.log" | grep -A20 "^----"
echo "== precise attitude"; $T --steps 600 -- --env-kw compensated_state=attitude 2>&1 | tee "$out/tune_attitude.log" | grep -A20 "^----"
)
;;
misc)
(
# Round-3 session: (1) fixed cost of a timed region of the headline bench (ms vs steps), (2) C3 / 2^18 with one queue + graph replay
# against the two-queue default, (3) the N sweep with and without graph replay.   usage: tools/r03_misc.sh <tag>
set -u
tag="${1:-misc1}"
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r03/$tag"; mkdir -p "$out"; cd "$root"
echo "== region time vs steps (two queues, 2^20 envs)"
for k in 10 20 40 80 160 320 640; do
  python bench.py --steps $k --warmup 5 --repeats 7 --no-cpu-baseline --no-sub-records 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('steps $k  region %.1f us  per step %.2f us (min %.2f max %.2f) wall %.2f' % (j['ms_per_step']*1e3*$k, j['ms_per_step']*1e3, j['ms_per_step_min_max'][0]*1e3, j['ms_per_step_min_max'][1]*1e3, j['wall_ms_per_step']*1e3))"
done | tee "$out/region_vs_steps.txt"
echo "== C3 (2^18): default (two queues) / one queue + graph / one queue plain"
for q in "" "--queues 1" "--queues 1 --env-kw rollout_graph=-1"; do
  python bench.py --workload c3 --envs-per-gpu 262144 --steps 2000 --no-cpu-baseline --no-sub-records $q 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('c3 [$q]  %.2f us/step (min %.2f max %.2f)' % (j['ms_per_step']*1e3, j['ms_per_step_min_max'][0]*1e3, j['ms_per_step_min_max'][1]*1e3))"
done | tee "$out/c3_queues.txt"
echo "== N sweep, bare C4 step: graph replay on / off"
for e in 10 12 14 16 17; do for g in 0 -1; do
  python bench.py --envs-per-gpu $((1<<e)) --steps 1000 --no-cpu-baseline --no-episode-stats --no-sub-records --env-kw rollout_graph=$g 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('N=2^$e graph=$g  %.2f us/step  %.3e env-steps/s' % (j['ms_per_step']*1e3, j['value']))"
done; done | tee "$out/n_sweep_graph.txt"
)
;;
final)
(
# Round-3 session on the final tree: the drift gates with their printed numbers, the new GPU tests, and the feature tiers against the
# exported round-2 tree (tools/r02_tree, see tools/tune.py --tree).   usage: tools/r03_final.sh <tag>
set -u
tag="${1:-final1}"
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r03/$tag"; mkdir -p "$out"; cd "$root"
python -m pytest tests/test_parity_gpu.py -q -m gpu -s -k "drift" 2>&1 | grep -E "drift|passed|failed" | tee "$out/drift_gates.txt"
python -m pytest tests/test_properties_gpu.py -q -m gpu -k "non_finite or graph or fresh" 2>&1 | tail -2
T="timeout -k 10 600 python tools/tune.py --rounds 2 --arms r02,r03 --tree r02=tools/r02_tree"
echo "== default make() tier (info_level=full), rollout"; $T --steps 1000 -- --env-kw info_level=full 2>&1 | tee "$out/tune_info.log" | grep -A4 "^----"
echo "== default make() tier, Python step loop (env-steps/s)"; $T --steps 1000 --field python_step_api_value -- --python-loop --env-kw info_level=full 2>&1 | tee "$out/tune_info_python.log" | grep -A4 "^----"
echo "== lean tier, Python step loop (env-steps/s)"; $T --steps 1000 --field python_step_api_value -- --python-loop 2>&1 | tee "$out/tune_lean_python.log" | grep -A4 "^----"
echo "== rl_reward + obs_goal_horizon=1"; $T --steps 1000 -- --env-kw cost=rl_reward --env-kw obs_goal_horizon=1 2>&1 | tee "$out/tune_goal_horizon.log" | grep -A4 "^----"
echo "== rl_reward + obs_goal_horizon=1, Python step loop, fresh outputs"; $T --steps 1000 --field python_step_api_value -- --python-loop --env-kw cost=rl_reward --env-kw obs_goal_horizon=1 --env-kw info_level=full 2>&1 | tee "$out/tune_goal_horizon_python.log" | grep -A4 "^----"
echo "== normalised actions"; $T --steps 1000 -- --env-kw normalized_rl_action_space=true --env-kw cost=rl_reward 2>&1 | tee "$out/tune_normalized.log" | grep -A4 "^----"
)
;;
lds_ab)
(
# A/B on one box, interleaved: X_GOAL gathered from global memory (ref_in_lds=0) vs staged in LDS (1), goal-horizon RL configuration.
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$root"; mkdir -p gpurun_out/r03
for r in 1 2 3; do for v in 0 1; do
  python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-sub-records --env-kw cost=rl_reward --env-kw obs_goal_horizon=1 --env-kw ref_in_lds=$v 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('round $r ref_in_lds=$v  %.2f us/step' % (j['ms_per_step']*1e3))"
done; done | tee gpurun_out/r03/goal_horizon_ref_in_lds_ab.txt
for r in 1 2; do for v in 0 1; do
  python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-sub-records --env-kw ref_in_lds=$v 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('headline round $r ref_in_lds=$v  %.2f us/step' % (j['ms_per_step']*1e3))"
done; done | tee -a gpurun_out/r03/goal_horizon_ref_in_lds_ab.txt
)
;;
close)
(
# Closing session of round 3: the full GPU suite (log kept under profiles/r03/), a one-off large configuration fuzz, the driver's command.
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$root"; out=gpurun_out/r03/close; mkdir -p $out
python -m pytest tests -m gpu -q -rA -s > $out/gpu_tests.log 2>&1; tail -1 $out/gpu_tests.log
SCG_FUZZ_SEEDS=${FUZZ:-400} timeout -k 10 900 python -m pytest tests/test_config_fuzz_gpu.py -m gpu -q > $out/gpu_config_fuzz_big.log 2>&1; tail -1 $out/gpu_config_fuzz_big.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.err; cut -c1-330 $out/bench_driver_cmd.json
python -c "import __graft_entry__ as g; g.smoke()"
)
;;
residuals)
(
# attitude precise mode: int16 residual planes (compensated_state='attitude') vs float32 ones ('attitude_f32'), interleaved on one box
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$root"; out=gpurun_out/r03/residuals; mkdir -p $out
python -m pytest tests/test_parity_gpu.py tests/test_config_fuzz_gpu.py -q -m gpu -s -k "precise or drift or fuzz or random_configuration" 2>&1 | grep -E "drift|passed|failed|FAILED|Error" | tee $out/tests.txt
for r in 1 2 3; do for m in attitude attitude_f32; do
  python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-sub-records --no-episode-stats --env-kw compensated_state=$m 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('round $r %-13s %.2f us/step  %d B moved  %.3f of peak' % ('$m', j['ms_per_step']*1e3, j['config']['moved_bytes_per_env_step'], j['roofline']['frac']))"
done; done | tee $out/attitude_int16_vs_f32.txt
)
;;
*) echo "usage: tools/r03_sessions.sh ab|tiers|misc|final|lds_ab|close|residuals [args]"; exit 2 ;;
esac
