#!/bin/bash
# C3 (Quadrotor-2D, 2^18 envs) and the bare C4 step at 2^18 / 2^19 envs: two launch queues against one queue (+ the in-library graph replay),
# interleaved on ONE box -- the driver's boxes put C3 anywhere between 7.6 and 10.1 us per step with the two-queue default.
set -e
out=gpurun_out/r04/c3_queues; mkdir -p $out
for rep in 1 2 3; do
  for q in 2 1; do
    python bench.py --workload c3 --envs-per-gpu 262144 --steps 1000 --warmup 100 --queues $q --no-cpu-baseline --no-sub-records > $out/c3_q${q}_r${rep}.json 2>> $out/stderr.log
    python - $out/c3_q${q}_r${rep}.json "C3 2^18 queues=$q rep $rep" <<'P' >> $out/summary.txt
import json, sys
d = json.load(open(sys.argv[1])); print('%s: %.2f us per step (min %.2f max %.2f)' % (sys.argv[2], d['ms_per_step'] * 1e3, d['ms_per_step_min_max'][0] * 1e3, d['ms_per_step_min_max'][1] * 1e3))
P
  done
done
for n in 262144 524288; do
  for q in 2 1; do
    python bench.py --envs-per-gpu $n --no-episode-stats --steps 1000 --warmup 100 --queues $q --no-cpu-baseline --no-sub-records > $out/c4_${n}_q${q}.json 2>> $out/stderr.log
    python - $out/c4_${n}_q${q}.json "bare C4 $n envs queues=$q" <<'P' >> $out/summary.txt
import json, sys
d = json.load(open(sys.argv[1])); print('%s: %.2f us per step (min %.2f max %.2f)' % (sys.argv[2], d['ms_per_step'] * 1e3, d['ms_per_step_min_max'][0] * 1e3, d['ms_per_step_min_max'][1] * 1e3))
P
  done
done
cat $out/summary.txt
