#!/usr/bin/env python3
"""Per-launch averages of every counter in the rocprofv3 --pmc passes below a directory, for kernels whose name contains a pattern.
usage: pmc_summary.py <dir> <kernel name pattern>"""
import collections
import csv
import glob
import sys

out, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if pat not in row.get('Kernel_Name', ''):
            continue
        k = row['Counter_Name']
        acc[k][0] += float(row['Counter_Value'])
        acc[k][1] += 1
for k, (s, c) in sorted(acc.items()):
    print('%-28s per-launch avg %.6g   (launches %d)' % (k, s / max(c, 1), c))
if not acc:
    print('no counter rows for kernels matching %r under %s' % (pat, out))
