#!/bin/bash
# the normalisers' calls by host + device clock (tools/bench_normalize.py), and the tests that pin them
cd "$(dirname "$0")/.." || exit 1
out="gpurun_out/r04/norm_host"; mkdir -p "$out"
timeout -k 10 400 python -m pytest tests/test_normalization.py tests/test_gae.py tests/test_examples_gpu.py -m gpu -q -x 2>&1 | tail -3 || exit 1
for r in 1 2 3; do timeout -k 10 120 python tools/bench_normalize.py 12 1048576; done 2>&1 | tee "$out/bench_normalize.txt"
for r in 1 2; do timeout -k 10 120 python tools/bench_normalize.py 24 16384; done 2>&1 | tee -a "$out/bench_normalize.txt"
timeout -k 10 600 python tools/rl_pipeline_only.py 2>/dev/null | tee "$out/bench_rl_pipeline.json" | cut -c1-1500
