#!/bin/bash
# the headline workload (C4 + episode returns, scg_rollout on two queues) beyond BASELINE's 2^20 envs per GPU
cd "$(dirname "$0")/.." || exit 1
out="gpurun_out/r04/big_batches"; mkdir -p "$out"
for e in ${SIZES:-20 22 24 26 28}; do
  n=$((1 << e)); steps=$(( e >= 26 ? 60 : 300 ))
  timeout -k 10 300 python bench.py --envs-per-gpu $n --steps $steps --action-period $(( e >= 26 ? 4 : 300 )) --warmup 10 --repeats 3 --no-cpu-baseline --no-sub-records > "$out/bench_2p$e.json" 2> "$out/bench_2p$e.err" || { tail -3 "$out/bench_2p$e.err"; exit 1; }
  python - "$out/bench_2p$e.json" $e <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('2^%s envs: %.3f ms per step, %.2f G env-steps/s, frac %.3f' % (sys.argv[2], r['ms_per_step'], r['value']/1e9, r['roofline']['frac']))
PY
done 2>&1 | tee "$out/table.txt"
