#!/usr/bin/env python3
"""Two-queue timeline from a rocprofv3 --kernel-trace CSV: begin / end of ~10 consecutive steps of scg_step_kernel on both HIP
streams, the step period they show, and the per-kernel average.  usage: timeline_excerpt.py kernel_trace.csv out.txt"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if 'scg_step_kernel' not in r.get('Kernel_Name', ''):
            continue
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', r.get('Stream_Id', '?')), r.get('Grid_Size', r.get('Grid_Size_X', '?'))))
rows.sort()
out = []
if not rows:
    out.append('no scg_step_kernel dispatches in the trace')
else:
    queues = sorted({q for _, _, q, _ in rows})
    out.append('dispatches of scg_step_kernel: %d on queues %s' % (len(rows), queues))
    mid = len(rows) // 2
    t0 = rows[mid][0]
    out.append('excerpt: 20 consecutive dispatches from the middle of the run (times in us relative to the first one)')
    out.append('%8s %10s %10s %9s  %s' % ('queue', 'begin', 'end', 'dur', 'grid'))
    for b, e, q, g in rows[mid:mid + 20]:
        out.append('%8s %10.2f %10.2f %9.2f  %s' % (q, (b - t0) / 1e3, (e - t0) / 1e3, (e - b) / 1e3, g))
    durs = [(e - b) / 1e3 for b, e, _, _ in rows]
    out.append('kernel duration: mean %.2f us, min %.2f, max %.2f (n = %d)' % (sum(durs) / len(durs), min(durs), max(durs), len(durs)))
    for q in queues:
        qs = [(b, e) for b, e, qq, _ in rows if qq == q]
        # steady-state period on this queue: consecutive begin-to-begin gaps inside runs of back-to-back launches
        gaps = sorted((qs[i + 1][0] - qs[i][0]) / 1e3 for i in range(len(qs) - 1))
        gaps = [g for g in gaps if g < 100.0]
        if gaps:
            out.append('queue %s: %d dispatches, begin-to-begin period median %.2f us (p10 %.2f, p90 %.2f); each dispatch is one HALF batch, '
                       'so this is the period of a step of the whole batch' % (q, len(qs), gaps[len(gaps) // 2], gaps[len(gaps) // 10], gaps[len(gaps) * 9 // 10]))
    # overlap: fraction of the time two step kernels run at once
    ev = sorted([(b, 1) for b, _, _, _ in rows] + [(e, -1) for _, e, _, _ in rows])
    live, last, both, any_ = 0, ev[0][0], 0, 0
    for t, d in ev:
        if live >= 1: any_ += t - last
        if live >= 2: both += t - last
        live += d; last = t
    out.append('two step kernels in flight during %.1f %% of the time at least one is' % (100.0 * both / max(any_, 1)))
open(sys.argv[2], 'w').write('\n'.join(out) + '\n')
print('\n'.join(out[-4:]))
