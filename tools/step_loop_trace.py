#!/usr/bin/env python
"""Kernel durations and start-to-start intervals of the step kernels in a rocprofv3 kernel trace (csv): tells a slower kernel from a
longer gap between dependent launches.  usage: step_loop_trace.py <kernel_trace.csv>"""
import csv
import statistics
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'scg_step_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
gap = [(int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3 for a, b in zip(rows, rows[1:])]
period = [(int(b['Start_Timestamp']) - int(a['Start_Timestamp'])) / 1e3 for a, b in zip(rows, rows[1:])]
q = lambda v: 'median %.2f  p10 %.2f  p90 %.2f' % (statistics.median(v), sorted(v)[len(v) // 10], sorted(v)[len(v) * 9 // 10])
print('%d step kernels  duration: %s | end-to-start gap: %s | period: %s' % (len(rows), q(dur), q(gap), q(period)))
