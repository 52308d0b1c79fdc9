#!/bin/bash
# C3 in the form that ships since the split moved above 2^18 envs: one queue, the rollout's launches replayed from the library's graph.
# rocprofv3 kernel trace + stats: kernel duration against the node-to-node period.
set -e
cd /tmp 2>/dev/null && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04/c3_trace; mkdir -p $out
d=$out/trace; rm -rf $d
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --workload c3 --envs-per-gpu 262144 --steps 1500 --warmup 100 --repeats 3 --no-cpu-baseline --no-sub-records > $out/bench.json 2> $out/bench.err
f=$(find $d -name '*kernel_trace.csv' | head -1)
{ echo "== c3, library default (one queue, graph replay; split above 2^18)"; python tools/step_loop_trace.py "$f"; } | tee $out/c3_timeline_default.txt
python tools/timeline_excerpt.py "$f" $out/c3_timeline_excerpt_default.txt > /dev/null
s=$(find $d -name '*kernel_stats.csv' | head -1); [ -n "$s" ] && cp "$s" $out/c3_kernel_stats_default.csv
python -c "import json; j=json.loads([l for l in open('$out/bench.json') if l.startswith('{')][-1]); print('bench under the profiler: %.2f us/step (events), wall %.2f' % (j['ms_per_step']*1e3, j['wall_ms_per_step']*1e3))" | tee -a $out/c3_timeline_default.txt
rm -rf $d
