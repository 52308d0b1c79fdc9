#!/bin/bash
# A/B of the reward normaliser's kernels by rocprof kernel durations: norm_old.so (variant) against the shipped library
cd "$(dirname "$0")/.." || exit 1
root=$(pwd); out="$root/gpurun_out/r04/norm_ab"; mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_normalization.py tests/test_gae.py -m gpu -q -x 2>&1 | tail -3 || exit 1
for round in 1 2; do
  for arm in old new; do
    d="$out/trace_${arm}_$round"; rm -rf "$d"
    if [ "$arm" = old ]; then export SCG_ROLLOUT_LIB="$root/safe_control_gym_amd/lib/variants/norm_old.so"; else unset SCG_ROLLOUT_LIB; fi
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 tools/bench_normalize.py 12 1048576 > "$d.log" 2>&1
    echo "== $arm round $round: $(cat "$d.log" | tail -1)"
    f=$(find "$d" -name '*kernel_stats.csv' | head -1)
    grep -E "reward|moments|apply" "$f" | cut -d, -f1-5 | sed 's/^/   /'
  done
done 2>&1 | tee "$out/reward_normalizer_ab.txt"
