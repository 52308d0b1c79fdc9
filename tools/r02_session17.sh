#!/bin/bash
# round 2, GPU session 17: asynchronous all-gather (RCCL one-rank world + the 2-rank gloo rehearsal), full suite.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02q"
mkdir -p "$out"
cd "$root"
echo "== tests"; timeout -k 10 1000 python -m pytest tests -m gpu -q -rA -s > "$out/gpu_tests.log" 2>&1; rc=$?; tail -2 "$out/gpu_tests.log"
grep -E "^(FAILED|ERROR)" "$out/gpu_tests.log" | head -30
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "abnormal end rc=$rc"; exit $rc; fi
echo "== 2-rank rehearsal on one GPU (gloo)"
SCG_SINGLE_DEVICE=1 timeout -k 10 600 python bench.py --gpus 2 --steps 20 --warmup 5 --envs-per-gpu 262144 > "$out/bench_2rank_rehearsal.json" 2> "$out/bench_2rank_rehearsal.err" || tail -5 "$out/bench_2rank_rehearsal.err"
python -c "import json; j=json.load(open('$out/bench_2rank_rehearsal.json')); print(j['n_gpus'], j['backend'], j['ranks_seen'], '%.2f us/step' % (j['ms_per_step']*1e3), j['config']['sharding'])"
