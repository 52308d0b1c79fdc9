#!/bin/bash
# One GPU-box session that produces the measured evidence committed under profiles/<tag>/ (run via gpurun).
# usage: tools/round_measure.sh <tag>      (round 2: r02)
set -u
tag="${1:-r02}"
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/$tag"
mkdir -p "$out"
cd "$root"
say() { echo "== $*"; }
say "bench, the driver's command (--steps 20 --warmup 5)"; python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_cmd.json" 2> "$out/bench_driver_cmd.err"; cut -c1-400 "$out/bench_driver_cmd.json"
say "bench, defaults (3000 steps, 5 regions) + python loop + fused"; python bench.py --python-loop --fused > "$out/bench_c4.json" 2> "$out/bench_c4.err"; cut -c1-300 "$out/bench_c4.json"
say "bench c3 / c2 as primary workloads"; python bench.py --workload c3 --envs-per-gpu 262144 --steps 5000 --fused > "$out/bench_c3.json" 2>/dev/null; python bench.py --workload c2 --envs-per-gpu 65536 --steps 10000 --fused > "$out/bench_c2.json" 2>/dev/null
python - "$out" <<'PY'
import json, sys
for w in ('c3', 'c2'):
    j = json.load(open('%s/bench_%s.json' % (sys.argv[1], w)))
    print('%s: %.2f us/step frac %.3f; fused %.2f us/step' % (w, j['ms_per_step'] * 1e3, j['roofline']['frac'], j['fused_rollout']['us_per_step']))
PY
say "N sweep (bare step)"; for e in 10 12 14 16 18 20 22 23; do python bench.py --envs-per-gpu $((1<<e)) --steps 2000 --no-cpu-baseline --no-episode-stats --no-sub-records 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('N=2^$e  %.3e env-steps/s  %.2f us/step  frac %.3f' % (j['value'], j['roofline']['kernel_us_per_launch'], j['roofline']['frac']))"; done | tee "$out/n_sweep.txt"
say "one launch queue (same workload)"; python bench.py --queues 1 --no-cpu-baseline --no-sub-records > "$out/bench_c4_one_queue.json" 2>/dev/null; python -c "import json; j=json.load(open('$out/bench_c4_one_queue.json')); print('one queue: %.2f us/step frac %.3f' % (j['ms_per_step']*1e3, j['roofline']['frac']))"
say "2 ranks sharing this GPU over gloo, started by bench.py itself (rehearsal of the N > 1 path; not a scaling number)"
SCG_SINGLE_DEVICE=1 python bench.py --gpus 2 --steps 600 --warmup 50 --envs-per-gpu 524288 > "$out/bench_2rank_rehearsal.json" 2> "$out/bench_2rank.err"; cut -c1-700 "$out/bench_2rank_rehearsal.json"
say "next-row kernels"; python tools/bench_expm.py > "$out/bench_expm.json" 2>/dev/null; cat "$out/bench_expm.json" | cut -c1-260
python tools/bench_linearize.py 3 2>/dev/null | tail -1 > "$out/bench_linearize.json"; python tools/bench_linearize.py 2 2>/dev/null | tail -1 >> "$out/bench_linearize.json"; cat "$out/bench_linearize.json"
python tools/bench_normalize.py 12 2>/dev/null | tail -1 | tee "$out/bench_normalize.json"
python tools/bench_gae.py 2>/dev/null | tail -1 | tee "$out/bench_gae.json"
say "rocprof kernel trace of the driver's command and of a 300-step run (two launch queues: begin / end of every kernel)"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_driver" -- python3 "$root/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records > "$out/prof_driver.log" 2>&1 )
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -- python3 "$root/bench.py" --steps 300 --warmup 50 --repeats 3 --no-cpu-baseline --no-sub-records > "$out/prof.log" 2>&1 )
find "$out/prof_driver" -name "*kernel_stats.csv" -exec cp {} "$out/c4_driver_cmd_kernel_stats.csv" \;
find "$out/prof" -name "*kernel_stats.csv" -exec cp {} "$out/c4_kernel_stats.csv" \;
find "$out/prof" -name "*kernel_trace.csv" -exec cp {} "$out/c4_kernel_trace_full.csv" \;
head -3 "$out/c4_kernel_stats.csv" | cut -c1-220
python3 tools/timeline_excerpt.py "$out/c4_kernel_trace_full.csv" "$out/c4_two_queue_timeline.txt" && tail -8 "$out/c4_two_queue_timeline.txt"
rm -f "$out/c4_kernel_trace_full.csv"; rm -rf "$out/prof" "$out/prof_driver"
say "pmc, shipped kernel (default workload: all counter groups)"; profiles/pmc.sh "$out/pmc" > "$out/pmc_summary.txt" 2>&1; cat "$out/pmc_summary.txt"
say "pmc, shipped kernel, bare C4 step (traffic counters)"; PMC_TRAFFIC_ONLY=1 profiles/pmc.sh "$out/pmc_bare" --no-episode-stats > "$out/pmc_summary_bare.txt" 2>&1; cat "$out/pmc_summary_bare.txt"
if [ -f "$root/safe_control_gym_amd/lib/variants/r1kernel.so" ]; then
  say "pmc, round-1 kernel body (per-wave reset, all out-of-bound compares) for the before / after instruction counts"
  SCG_ROLLOUT_LIB="$root/safe_control_gym_amd/lib/variants/r1kernel.so" PMC_GROUPS_ONLY="SQ" profiles/pmc.sh "$out/pmc_r1kernel" > "$out/pmc_summary_r1kernel.txt" 2>&1; cat "$out/pmc_summary_r1kernel.txt"
fi
rm -rf "$out"/pmc/pass* "$out"/pmc_bare/pass* "$out"/pmc_r1kernel/pass*
