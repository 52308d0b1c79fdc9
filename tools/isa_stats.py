#!/usr/bin/env python3
"""Static instruction mix of one kernel in a `hipcc -S --offload-device-only` listing.

usage: tools/isa_stats.py kernels.s '<mangled-name substring>' [--dump out.s]
Counts are static (every instruction once, whatever the control flow); the PMC counter SQ_INSTS_VALU is the dynamic twin.
"""
import collections
import re
import sys


def body_of(path, key):
    lines = open(path).read().split('\n')
    start = None
    for i, l in enumerate(lines):
        if l.startswith('_ZN') and key in l and l.rstrip().split(':')[0].endswith('E') and ': ' in l or (l.startswith('_ZN') and key in l and l.split(':')[0] and l.strip().endswith(':')):
            start = i
            break
    if start is None:
        raise SystemExit('kernel not found: ' + key)
    out = []
    for l in lines[start + 1:]:
        if l.startswith('.Lfunc_end'):
            break
        out.append(l)
    return out


def main():
    path, key = sys.argv[1], sys.argv[2]
    body = body_of(path, key)
    if '--dump' in sys.argv:
        open(sys.argv[sys.argv.index('--dump') + 1], 'w').write('\n'.join(body))
    ops = collections.Counter()
    for l in body:
        l = l.strip()
        if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'):
            continue
        ops[l.split()[0]] += 1
    cls = collections.Counter()
    for op, c in ops.items():
        if op.startswith('v_'):
            cls['VALU'] += c
        elif op.startswith('s_'):
            if op.startswith('s_load') or op.startswith('s_buffer'):
                cls['SMEM'] += c
            elif op in ('s_waitcnt', 's_nop', 's_barrier'):
                cls[op] += c
            else:
                cls['SALU'] += c
        elif op.startswith('global_') or op.startswith('flat_') or op.startswith('buffer_') or op.startswith('scratch_'):
            cls['VMEM'] += c
        elif op.startswith('ds_'):
            cls['LDS'] += c
        else:
            cls['other'] += c
    print('class:', dict(cls))
    meta = [l for l in open(path).read().split('\n') if False]
    top = sorted(((c, op) for op, c in ops.items() if op.startswith('v_')), reverse=True)
    print('VALU by opcode:')
    for c, op in top[:60]:
        print('  %5d %s' % (c, op))


if __name__ == '__main__':
    main()
