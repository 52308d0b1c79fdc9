#!/bin/bash
# The GPU-box sessions of round 4 that produced the files under profiles/r04/ (one sub-command = one gpurun call = one box).
#   tools/r04_sessions.sh first <tag>      the new GPU tests, then the whole GPU suite, the step-API probe, the driver's command
#   tools/r04_sessions.sh lone <tag>       the lone dependent launch: kernel-trace duration (median / p90 of 1500 launches) and SQ
#                                          counters per arm -- the shipped library, each round-3 head / tail change reverted singly
#                                          (lib/variants, tools/build_variants.sh), the exported round-2 tree -- and the host-side A/B
#                                          (per-step torch.empty against a ring of pre-allocated output sets)
#   tools/r04_sessions.sh evidence <tag>   rocprofv3 kernel traces of the C3 / C2 workloads, SQ counters of C3, kernel stats of the
#                                          next-row kernels (GAE, advantage normalisation, normalisers, linearize)
#   tools/r04_sessions.sh probe <tag> ...  tools/step_api_probe.py with the given arguments
cmd="${1:-}"; shift || true
tag="${1:-run}"; shift || true
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r04/$tag"; mkdir -p "$out"; cd "$root"
export TMPDIR=/tmp
trace_of() { # <dir> -> prints the step-kernel duration line of the newest kernel trace below it
  f=$(find "$1" -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python tools/step_loop_trace.py "$f"
}
case "$cmd" in
first)
  echo "== new GPU tests"; timeout -k 10 900 python -m pytest tests/test_pipelines_gpu.py -m gpu -q -x > "$out/gpu_tests_new.log" 2>&1; tail -15 "$out/gpu_tests_new.log"
  echo "== step API probe"; timeout -k 10 600 python tools/step_api_probe.py --steps 400 --rounds 2 2>&1 | tee "$out/step_api_probe.txt" | tail -30
  echo "== GPU suite"; timeout -k 10 1500 python -m pytest tests -m gpu -q > "$out/gpu_tests.log" 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" "$out/gpu_tests.log" | tail -15
  echo "== driver's command"; timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_cmd.json" 2> "$out/bench_driver_cmd.err"; tail -c 600 "$out/bench_driver_cmd.json"
  ;;
second)
  echo "== new GPU tests"; timeout -k 10 900 python -m pytest tests/test_pipelines_gpu.py tests/test_gae.py tests/test_randomization.py -m gpu -q -x > "$out/gpu_tests_new.log" 2>&1; tail -15 "$out/gpu_tests_new.log"
  echo "== driver's command"; timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_cmd.json" 2> "$out/bench_driver_cmd.err"; tail -5 "$out/bench_driver_cmd.err"
  python - "$out/bench_driver_cmd.json" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('value %.4g  ms_per_step %.5f  frac %.3f' % (j['value'], j['ms_per_step'], j['roofline']['frac']))
for k in ('step_api', 'rl_pipeline', 'c3', 'c2'):
    print(k, json.dumps(j.get(k))[:1500])
for pt in j['n_sweep']['points']:
    print(pt)
PY
  for t in gae; do
    d="$out/trace_$t"; rm -rf "$d"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 tools/bench_$t.py > "$out/bench_$t.txt" 2> "$d.err"
    s=$(find "$d" -name '*kernel_stats.csv' | head -1); [ -n "$s" ] && cp "$s" "$out/${t}_kernel_stats.csv" && grep scg_ "$s" | cut -c1-200
    tail -3 "$out/bench_$t.txt"; rm -rf "$d"
  done
  ;;
fuzz)
  echo "== configuration fuzz (rollout / sink / pipelines == step loop)"; timeout -k 10 1100 python -m pytest tests/test_config_fuzz_gpu.py tests/test_examples_gpu.py -m gpu -q -x -k "rollout_equals or example" > "$out/gpu_fuzz.log" 2>&1; tail -12 "$out/gpu_fuzz.log"
  for args in "4096 128 planes" "4096 128 graph" "16384 128 pushed" "16384 128 planes" "16384 128 graph" "262144 128 pushed" "262144 128 planes" "262144 128 graph" "1048576 64 pushed" "1048576 64 planes" "1048576 64 graph"; do
    timeout -k 10 300 python examples/ppo_rollout.py $args 2>&1 | grep "env-steps/s" | tee -a "$out/ppo_rollout_example.txt"
  done
  ;;
close)
  # the closing session: whole GPU suite, the canonical evidence of the round (tools/round_measure.sh), traffic counters of the
  # trajectory-sink rollout, smoke()
  echo "== GPU suite"; timeout -k 10 1500 python -m pytest tests -m gpu -q > "$out/gpu_tests.log" 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" "$out/gpu_tests.log" | tail -15
  timeout -k 10 1500 bash tools/round_measure.sh "r04/$tag" 2>&1 | cut -c1-600
  echo "== pmc, trajectory-sink rollout, bare C4 step (traffic counters)"; PMC_TRAFFIC_ONLY=1 profiles/pmc.sh "$out/pmc_sink" --no-episode-stats --trajectory-sink 60 > "$out/pmc_summary_bare_trajectory_sink.txt" 2>&1; cat "$out/pmc_summary_bare_trajectory_sink.txt"; rm -rf "$out"/pmc_sink/pass*
  echo "== smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  ;;
final)
  # the last word of the round on the final tree: whole GPU suite, the driver's command, smoke()
  echo "== GPU suite"; timeout -k 10 1500 python -m pytest tests -m gpu -q > "$out/gpu_tests.log" 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" "$out/gpu_tests.log" | tail -15
  echo "== driver's command"; timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_cmd.json" 2> "$out/bench_driver_cmd.err"; cut -c1-300 "$out/bench_driver_cmd.json"
  echo "== smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  ;;
suite)
  echo "== GPU suite"; timeout -k 10 1500 python -m pytest tests -m gpu -q > "$out/gpu_tests.log" 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" "$out/gpu_tests.log" | tail -15
  echo "== smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  ;;
probe)
  timeout -k 10 900 python tools/step_api_probe.py "$@" 2>&1 | tee "$out/step_api_probe.txt"
  ;;
lone)
  arms="${ARMS:-shipped noserial trig_r02 perwave_reset stagger32 stagger64 stagger127}"
  tbl="$out/lone_launch_arms.txt"; : > "$tbl"
  for round in 1 2; do
    for arm in $arms r02; do
      d="$out/trace_${arm}_$round"; rm -rf "$d"
      if [ "$arm" = "r02" ]; then
        ( cd tools/r02_tree && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$d" -- python3 ../step_loop_only.py 1500 lean > "$d.log" 2>&1 )
      elif [ "$arm" = "shipped" ]; then
        timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$d" -- python3 tools/step_loop_only.py 1500 lean > "$d.log" 2>&1
      else
        SCG_ROLLOUT_LIB="$root/safe_control_gym_amd/lib/variants/$arm.so" timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$d" -- python3 tools/step_loop_only.py 1500 lean > "$d.log" 2>&1
      fi
      echo "lean    round $round $arm: $(trace_of "$d")" | tee -a "$tbl"
      rm -rf "$d"
    done
    # the default make() tier (fresh outputs): shipped / plain terminal-record stores / a ring of 3 and 4 output sets
    for arm in shipped nt_record0 ring3 ring4 r02; do
      d="$out/trace_default_${arm}_$round"; rm -rf "$d"
      case "$arm" in
        r02) ( cd tools/r02_tree && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$d" -- python3 ../step_loop_only.py 1500 default > "$d.log" 2>&1 ) ;;
        shipped) timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$d" -- python3 tools/step_loop_only.py 1500 default > "$d.log" 2>&1 ;;
        ring3) timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$d" -- python3 tools/step_loop_only.py 1500 default 3 > "$d.log" 2>&1 ;;
        ring4) timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$d" -- python3 tools/step_loop_only.py 1500 default 4 > "$d.log" 2>&1 ;;
        *) SCG_ROLLOUT_LIB="$root/safe_control_gym_amd/lib/variants/$arm.so" timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$d" -- python3 tools/step_loop_only.py 1500 default > "$d.log" 2>&1 ;;
      esac
      echo "default round $round $arm: $(trace_of "$d")" | tee -a "$tbl"
      rm -rf "$d"
    done
  done
  # SQ counters of the lone launch per arm (their own runs: counters serialise the kernels)
  for arm in $arms; do
    d="$out/pmc_$arm"; rm -rf "$d"
    lib=""; [ "$arm" != "shipped" ] && lib="$root/safe_control_gym_amd/lib/variants/$arm.so"
    for group in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
      g=$(echo $group | tr ' ' '_' | cut -c1-40)
      if [ -n "$lib" ]; then SCG_ROLLOUT_LIB="$lib" timeout -k 10 300 rocprofv3 --pmc $group --output-format csv -d "$d/$g" -- python3 tools/step_loop_only.py 200 lean > "$d.$g.log" 2>&1
      else timeout -k 10 300 rocprofv3 --pmc $group --output-format csv -d "$d/$g" -- python3 tools/step_loop_only.py 200 lean > "$d.$g.log" 2>&1; fi
    done
    python tools/pmc_summary.py "$d" scg_step_kernel > "$out/pmc_lone_$arm.txt" 2>&1; cat "$out/pmc_lone_$arm.txt"
    rm -rf "$d"
  done
  ;;
evidence)
  for w in c3 c2; do
    n=$([ $w = c3 ] && echo 262144 || echo 65536)
    for q in "" "--env-kw rollout_graph=-1"; do
      d="$out/trace_${w}_$(echo "$q" | tr -cd 'a-z0-9' | cut -c1-12)"; rm -rf "$d"
      timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 bench.py --workload $w --envs-per-gpu $n --steps 1500 --warmup 100 --repeats 1 --no-cpu-baseline --no-sub-records $q > "$d.json" 2> "$d.err"
      f=$(find "$d" -name '*kernel_trace.csv' | head -1)
      echo "== $w [$q]" | tee -a "$out/${w}_timeline.txt"
      python tools/step_loop_trace.py "$f" | tee -a "$out/${w}_timeline.txt"
      python tools/timeline_excerpt.py "$f" "$out/${w}_timeline_excerpt_$(echo "$q" | tr -cd 'a-z0-9' | cut -c1-12).txt" > /dev/null
      s=$(find "$d" -name '*kernel_stats.csv' | head -1); [ -n "$s" ] && cp "$s" "$out/${w}_kernel_stats_$(echo "$q" | tr -cd 'a-z0-9' | cut -c1-12).csv"
      python -c "import json,sys; j=json.loads([l for l in open('$d.json') if l.startswith('{')][-1]); print('bench: %.2f us/step (events), wall %.2f' % (j['ms_per_step']*1e3, j['wall_ms_per_step']*1e3))" | tee -a "$out/${w}_timeline.txt"
      rm -rf "$d"
    done
  done
  d="$out/pmc_c3"; rm -rf "$d"
  for group in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    g=$(echo $group | tr ' ' '_' | cut -c1-40)
    timeout -k 10 300 rocprofv3 --pmc $group --output-format csv -d "$d/$g" -- python3 bench.py --workload c3 --envs-per-gpu 262144 --steps 100 --warmup 10 --repeats 1 --no-cpu-baseline --no-sub-records > "$d.$g.log" 2>&1
  done
  python tools/pmc_summary.py "$d" scg_step_kernel | tee "$out/pmc_summary_c3.txt"; rm -rf "$d"
  for t in gae advnorm normalize linearize; do
    d="$out/trace_$t"; rm -rf "$d"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 tools/bench_$t.py > "$out/bench_$t.txt" 2> "$d.err"
    s=$(find "$d" -name '*kernel_stats.csv' | head -1); [ -n "$s" ] && cp "$s" "$out/${t}_kernel_stats.csv" && head -6 "$s"
    tail -3 "$out/bench_$t.txt"; rm -rf "$d"
  done
  ;;
loneab)
  # the lone-launch kernel forms (scg_config.lone_launch_forms: per-wave reset, plain record stores) on / off: Python step loops by
  # kernel trace, then the queued forms must not have moved: headline two-queue rollout, default tier rollout, C3 / C2 single queue
  tbl="$out/lone_forms_ab.txt"; : > "$tbl"
  for round in 1 2; do
    for tier in lean default; do
      for forms in 0 -1; do
        d="$out/trace_${tier}_${forms}_$round"; rm -rf "$d"
        SCG_LONE_LAUNCH_FORMS=$forms timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$d" -- python3 tools/step_loop_only.py 1500 $tier > "$d.log" 2>&1
        echo "$tier round $round lone_launch_forms=$forms: $(trace_of "$d")" | tee -a "$tbl"
        rm -rf "$d"
      done
    done
  done
  B="python bench.py --no-cpu-baseline --no-sub-records"
  pr() { python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%s  %.2f us/step (min %.2f max %.2f)' % ('$1', j['ms_per_step']*1e3, j['ms_per_step_min_max'][0]*1e3, j['ms_per_step_min_max'][1]*1e3))"; }
  for round in 1 2; do
    $B --steps 1500 2>/dev/null | pr "headline two queues" | tee -a "$tbl"
    $B --steps 1000 --env-kw info_level=full 2>/dev/null | pr "default tier two queues" | tee -a "$tbl"
    for f in 0 1; do
      $B --steps 1500 --queues 1 --env-kw lone_launch_forms=$f 2>/dev/null | pr "C4 one queue lone_launch_forms=$f" | tee -a "$tbl"
      $B --steps 1000 --queues 1 --env-kw info_level=full --env-kw lone_launch_forms=$f 2>/dev/null | pr "default tier one queue lone_launch_forms=$f" | tee -a "$tbl"
      $B --workload c3 --envs-per-gpu 262144 --steps 2000 --queues 1 --env-kw lone_launch_forms=$f 2>/dev/null | pr "C3 one queue lone_launch_forms=$f" | tee -a "$tbl"
      $B --workload c2 --envs-per-gpu 65536 --steps 2000 --env-kw lone_launch_forms=$f 2>/dev/null | pr "C2 lone_launch_forms=$f" | tee -a "$tbl"
    done
  done
  ;;
timeline)
  # kernel timelines of the policy-in-the-loop step loop: plain, two / three shards on their own queues, two shards joined to the
  # caller's stream by events; then the same loops replayed from a HIP graph (no host in the loop)
  tl="$out/step_api_timeline.txt"; : > "$tl"
  for g in 0 50; do
    for mode in plain pipe2_queue pipe3_queue pipe2; do
      d="$out/trace_${mode}_g$g"; rm -rf "$d"
      timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$d" -- python3 tools/step_api_probe.py --tiers lean --modes $mode --steps 400 --rounds 1 --graph $g > "$d.log" 2>&1
      f=$(find "$d" -name '*kernel_trace.csv' | head -1)
      grep "^round" "$d.log" | tee -a "$tl"
      [ -n "$f" ] && python tools/pipeline_timeline.py "$f" "$tl" "lean tier, 2^20 envs, $mode$([ $g != 0 ] && echo ', replayed from a HIP graph of 50 steps')"
      rm -rf "$d"
    done
  done
  ;;
*) echo "usage: $0 first|probe|lone|evidence|timeline <tag>"; exit 2 ;;
esac
