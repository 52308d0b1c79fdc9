#!/bin/bash
# one-launch (small batch) against two-launch normaliser calls: tools/bench_normalize.py at several batch sizes, interleaved, then
# the kernel durations of both forms by rocprofv3 kernel stats
cd "$(dirname "$0")/.." || exit 1
root=$(pwd); out="$root/gpurun_out/r04/norm_small"; mkdir -p "$out"; export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_normalization.py tests/test_gae.py tests/test_reference_pins.py -m gpu -q -x -k "normal or gae or ppo" 2>&1 | tail -3 || exit 1
two="$root/safe_control_gym_amd/lib/variants/norm_two_launches.so"
for N in 1024 4096 16384; do
  for arm in two one two one; do
    if [ "$arm" = two ]; then export SCG_ROLLOUT_LIB="$two"; else unset SCG_ROLLOUT_LIB; fi
    echo "N=$N $arm: $(timeout -k 10 120 python tools/bench_normalize.py 12 $N 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('mean_std %.2f us  reward %.2f us' % (r['us'], r['reward_us']))")"
  done
done 2>&1 | tee "$out/table.txt"
for N in 4096 16384; do
  for arm in two one; do
    if [ "$arm" = two ]; then export SCG_ROLLOUT_LIB="$two"; else unset SCG_ROLLOUT_LIB; fi
    d="$out/trace_${arm}_$N"; rm -rf "$d"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 tools/bench_normalize.py 12 $N > "$d.log" 2>&1
    f=$(find "$d" -name '*kernel_stats.csv' | head -1)
    echo "== N=$N $arm (rocprof kernel stats)"
    python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('reward','moments','apply','small')):
        print('   %-44s calls %s avg %.2f us min %.2f max %.2f'%(r['Name'].split('(')[0][-44:], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
    rm -rf "$d"
  done
done 2>&1 | tee -a "$out/table.txt"
