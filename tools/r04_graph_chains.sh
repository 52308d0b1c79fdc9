#!/bin/bash
# Rollouts replayed from a hipGraph with the batch as ONE chain of nodes against TWO independent chains (the halves of the batch as
# two branches of the graph; SCG_GRAPH_CHAINS_MIN = smallest half batch that takes two), interleaved on one box.  --queues 1 keeps the
# plain two-queue split out of the way at 2^18 and above.
set -e
out=gpurun_out/r04/graph_chains; mkdir -p $out
one() {  # label, chains_min, bench args...
  label=$1; min=$2; shift 2
  SCG_GRAPH_CHAINS_MIN=$min python bench.py "$@" --steps 1000 --warmup 100 --queues 1 --no-cpu-baseline --no-sub-records > $out/tmp.json 2>> $out/stderr.log
  python - $out/tmp.json "$label" <<'P' >> $out/summary.txt
import json, sys
d = json.load(open(sys.argv[1])); print('%s: %.2f us per step (min %.2f max %.2f)' % (sys.argv[2], d['ms_per_step'] * 1e3, d['ms_per_step_min_max'][0] * 1e3, d['ms_per_step_min_max'][1] * 1e3))
P
}
for rep in 1 2; do
  for min in 0 1; do
    one "C2 2^16 chains_min=$min rep $rep" $min --workload c2 --envs-per-gpu 65536
    one "C3 2^18 chains_min=$min rep $rep" $min --workload c3 --envs-per-gpu 262144
    for n in 4096 16384 65536 262144 524288; do
      one "bare C4 $n envs chains_min=$min rep $rep" $min --envs-per-gpu $n --no-episode-stats
    done
  done
done
cat $out/summary.txt
