#!/bin/bash
# round 2, GPU session 23: non-finite inputs; bench with the three CPU baselines.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02v"
mkdir -p "$out"
cd "$root"
timeout -k 10 600 python -m pytest tests/test_properties_gpu.py -m gpu -q -rA -s -k "non_finite" > "$out/nonfinite.log" 2>&1; tail -3 "$out/nonfinite.log"; grep -n "^E  " "$out/nonfinite.log" | head -10
python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_cmd.json" 2> "$out/bench_driver_cmd.err" || tail -3 "$out/bench_driver_cmd.err"
python -c "import json; j=json.load(open('$out/bench_driver_cmd.json')); print('%.2f us'%(j['ms_per_step']*1e3), j['cpu_baseline'])"
