#!/usr/bin/env python3
"""Timeline of a policy-in-the-loop step loop from a rocprofv3 --kernel-trace CSV: per launch queue the chain
policy kernel -> step kernel -> policy kernel ..., with kernel durations, the idle gaps between dependent kernels of a queue,
the period of a step of the whole batch, and how much of the time two or more kernels are in flight.
usage: pipeline_timeline.py kernel_trace.csv out.txt [label]"""
import csv
import statistics
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r.get('Kernel_Name', '')
        kind = 'step' if 'scg_step_kernel' in name else ('reset' if 'scg_reset' in name else 'policy/other')
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', r.get('Stream_Id', '?')), kind, name[:60]))
rows.sort()
label = sys.argv[3] if len(sys.argv) > 3 else ''
out = ['== %s' % label] if label else []
steps = [r for r in rows if r[3] == 'step']
if len(steps) < 40:
    out.append('fewer than 40 step kernels in the trace')
else:
    # the steady state: the middle half of the run
    t_lo, t_hi = steps[len(steps) // 4][0], steps[3 * len(steps) // 4][0]
    mid = [r for r in rows if t_lo <= r[0] < t_hi]
    queues = sorted({r[2] for r in mid})
    out.append('kernels in the middle half of the run: %d on queues %s' % (len(mid), queues))
    t0 = mid[0][0]
    out.append('excerpt (us relative to its first kernel):')
    out.append('%8s %-13s %9s %9s %8s' % ('queue', 'kernel', 'begin', 'end', 'dur'))
    for b, e, q, kind, _ in mid[:28]:
        out.append('%8s %-13s %9.2f %9.2f %8.2f' % (q, kind, (b - t0) / 1e3, (e - t0) / 1e3, (e - b) / 1e3))
    med = lambda v: statistics.median(v) if v else float('nan')
    n_step_mid = sum(1 for r in mid if r[3] == 'step')
    for q in queues:
        qs = [r for r in mid if r[2] == q]
        sd = [(e - b) / 1e3 for b, e, _, k, _ in qs if k == 'step']
        pd = [(e - b) / 1e3 for b, e, _, k, _ in qs if k != 'step']
        gaps = [(qs[i + 1][0] - qs[i][1]) / 1e3 for i in range(len(qs) - 1)]
        per = [(b2 - b1) / 1e3 for (b1, _, _, k1, _), (b2, _, _, k2, _) in zip([r for r in qs if r[3] == 'step'], [r for r in qs if r[3] == 'step'][1:])]
        out.append('queue %s: step kernel median %.2f us (n=%d), other kernels median %.2f us (n=%d), idle gap between consecutive kernels '
                   'median %.2f us (p90 %.2f), step-to-step period median %.2f us'
                   % (q, med(sd), len(sd), med(pd), len(pd), med(gaps), sorted(gaps)[len(gaps) * 9 // 10] if gaps else float('nan'), med(per)))
    shards = max(1, len({r[2] for r in mid if r[3] == 'step'}))
    out.append('step kernels per queue carrying steps: %d shard queue(s); whole-batch steps in the window: %.1f; window %.1f us -> %.2f us per step of the '
               'whole batch' % (shards, n_step_mid / shards, (t_hi - t_lo) / 1e3, (t_hi - t_lo) / 1e3 / (n_step_mid / shards)))
    ev = sorted([(b, 1) for b, _, _, _, _ in mid] + [(e, -1) for _, e, _, _, _ in mid])
    live, last, busy = 0, ev[0][0], [0, 0, 0, 0]
    for t, d in ev:
        busy[min(live, 3)] += t - last
        live += d
        last = t
    tot = sum(busy)
    out.append('kernels in flight: none %.1f %%, one %.1f %%, two %.1f %%, three or more %.1f %% of the window'
               % tuple(100.0 * x / max(tot, 1) for x in busy))
open(sys.argv[2], 'a').write('\n'.join(out) + '\n')
print('\n'.join(out[-6:]))
