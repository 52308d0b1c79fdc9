#!/bin/bash
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
