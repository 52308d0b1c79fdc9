#!/usr/bin/env python3
"""VALU instructions per basic block of a kernel body dumped by tools/isa_stats.py --dump (static counts)."""
import re
import sys
lines = open(sys.argv[1]).read().split('\n')
cur, c, out = 'entry', 0, []
for l in lines:
    s = l.strip()
    if re.match(r'^\.LBB\d+_\d+:', s):
        out.append((cur, c)); cur, c = s.split(':')[0], 0
    elif s.startswith('v_'):
        c += 1
out.append((cur, c))
print(' '.join('%s:%d' % (k.replace('.LBB', 'B'), v) for k, v in out if v > 0))
