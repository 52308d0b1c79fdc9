#!/bin/bash
# round 2, GPU session 35: the N > 1 code path of bench.py on RCCL with a forced one-rank world (every collective runs).
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02ae"
mkdir -p "$out"
cd "$root"
SCG_BENCH_FORCE_COLLECTIVES=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records > "$out/bench_forced_collectives.json" 2> "$out/bench_forced_collectives.err" || tail -8 "$out/bench_forced_collectives.err"
python -c "import json; j=json.loads([l for l in open('$out/bench_forced_collectives.json') if l.startswith('{')][-1]); print(j['n_gpus'], j['backend'], j['ranks_seen'], j['forced_collectives'], '%.2f us' % (j['ms_per_step']*1e3), j.get('per_rank_ms_per_step_min_max'))"
SCG_BENCH_FORCE_COLLECTIVES=1 timeout -k 10 300 python bench.py --gpus 1 --steps 900 --warmup 50 --repeats 3 --no-cpu-baseline --no-sub-records > "$out/bench_forced_collectives_900.json" 2>/dev/null
python -c "import json; j=json.loads([l for l in open('$out/bench_forced_collectives_900.json') if l.startswith('{')][-1]); print('900 steps (3 gathers per region): %.2f us' % (j['ms_per_step']*1e3))"
python bench.py --gpus 1 --steps 900 --warmup 50 --repeats 3 --no-cpu-baseline --no-sub-records 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('900 steps, no collectives: %.2f us' % (j['ms_per_step']*1e3))"
