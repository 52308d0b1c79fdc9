#!/bin/bash
# Right above the split threshold: two launch queues (default there) against one queue + graph replay, for the 2-D and the 3-D quadrotor.
set -e
out=gpurun_out/r04/split_range; mkdir -p $out; : > $out/summary.txt
for n in 294912 327680 393216; do
  for w in c3 c4; do
    for q in 2 1; do
      extra=$([ $w = c4 ] && echo --no-episode-stats || true)
      python bench.py --workload $w --envs-per-gpu $n $extra --steps 1000 --warmup 100 --queues $q --no-cpu-baseline --no-sub-records > $out/tmp.json 2>> $out/stderr.log
      python - $out/tmp.json "$w $n envs queues=$q" <<'P' >> $out/summary.txt
import json, sys
d = json.load(open(sys.argv[1])); print('%s: %.2f us per step (min %.2f max %.2f)' % (sys.argv[2], d['ms_per_step'] * 1e3, d['ms_per_step_min_max'][0] * 1e3, d['ms_per_step_min_max'][1] * 1e3))
P
    done
  done
done
cat $out/summary.txt
