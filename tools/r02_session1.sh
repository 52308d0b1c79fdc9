#!/bin/bash
# round 2, GPU session 1: the full GPU suite on the new ABI, the driver's bench command, and an A/B of the kernel variants.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02a"
mkdir -p "$out"
cd "$root"
echo "== tests"; timeout -k 10 900 python -m pytest tests -m gpu -q > "$out/gpu_tests.log" 2>&1; rc=$?; tail -5 "$out/gpu_tests.log"
grep -E "^(FAILED|ERROR)" "$out/gpu_tests.log" | head -20
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "test run ended abnormally (rc=$rc): stop"; exit $rc; fi
echo "== bench, the driver's command"; python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_cmd.json" 2> "$out/bench_driver_cmd.err" || { tail -20 "$out/bench_driver_cmd.err"; exit 1; }
cat "$out/bench_driver_cmd.json"
echo "== variants"; python tools/tune.py --rounds 3 --steps 1500 > "$out/tune_variants.log" 2>&1; tail -12 "$out/tune_variants.log"
echo "== variants, one launch queue"; python tools/tune.py --rounds 2 --steps 1500 -- --queues 1 > "$out/tune_variants_q1.log" 2>&1; tail -8 "$out/tune_variants_q1.log"
