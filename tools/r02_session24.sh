#!/bin/bash
# round 2, GPU session 24: full suite + the driver's command on the final tree.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02w"
mkdir -p "$out"
cd "$root"
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== tests"; timeout -k 10 1000 python -m pytest tests -m gpu -q -rA -s > "$out/gpu_tests.log" 2>&1; rc=$?; tail -2 "$out/gpu_tests.log"
grep -E "^(FAILED|ERROR)" "$out/gpu_tests.log" | head -30
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "abnormal end rc=$rc"; exit $rc; fi
echo "== bench, driver's command"; ( time python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_cmd.json" 2> "$out/bench_driver_cmd.err" ) 2>&1 | grep real
python -c "import json; j=json.load(open('$out/bench_driver_cmd.json')); print('%.2f us frac %.3f'%(j['ms_per_step']*1e3, j['roofline']['frac'])); print(j['cpu_baseline'])"
echo "== bench, defaults"; ( time python bench.py > "$out/bench_c4.json" 2> "$out/bench_c4.err" ) 2>&1 | grep real
python -c "import json; j=json.load(open('$out/bench_c4.json')); print('%.2f us frac %.3f'%(j['ms_per_step']*1e3, j['roofline']['frac']))"
