#!/bin/bash
# Closing session of round 3: the full GPU suite (log kept under profiles/r03/), a one-off large configuration fuzz, the driver's command.
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$root"; out=gpurun_out/r03/close; mkdir -p $out
python -m pytest tests -m gpu -q -rA -s > $out/gpu_tests.log 2>&1; tail -1 $out/gpu_tests.log
SCG_FUZZ_SEEDS=${FUZZ:-400} timeout -k 10 900 python -m pytest tests/test_config_fuzz_gpu.py -m gpu -q > $out/gpu_config_fuzz_big.log 2>&1; tail -1 $out/gpu_config_fuzz_big.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.err; cut -c1-330 $out/bench_driver_cmd.json
python -c "import __graft_entry__ as g; g.smoke()"
