#!/usr/bin/env python
"""A/B the kernel build variants and launch-geometry knobs on the GPU box (tuning only).

Each arm runs bench.py in a child process (fresh library load) and reports kernel us/launch; arms are
interleaved over `--rounds` rounds so that clock / thermal drift hits all arms alike.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(lib, env_extra, steps, extra_args, abi=None, tree=None):
    env = dict(os.environ)
    if lib:
        env['SCG_ROLLOUT_LIB'] = lib
    env.update(env_extra)
    bench = os.path.join(tree or ROOT, 'bench.py')
    argv = ['--steps', str(steps), '--warmup', '100', '--no-cpu-baseline', '--no-sub-records'] + extra_args
    if abi is None:
        cmd = [sys.executable, bench] + argv
    else:
        # an arm kept from an earlier ABI version whose structs have the same layout (e.g. a copy of last round's library): this
        # process alone tells the binding to expect that version -- nothing in the package has such a switch
        cmd = [sys.executable, '-c', 'import sys, runpy; sys.path.insert(0, %r); import safe_control_gym_amd._capi as c; '
               'c.SCG_ABI_VERSION = %d; sys.argv = [%r] + sys.argv[1:]; runpy.run_path(%r, run_name="__main__")'
               % (ROOT, int(abi), bench, bench)] + argv
    out = subprocess.run(cmd, env=env, capture_output=True, text=True)
    for line in out.stdout.splitlines():
        if line.startswith('{'):
            return json.loads(line)
    return {'error': out.stderr[-400:]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=2)
    ap.add_argument('--steps', type=int, default=1500)
    ap.add_argument('--arms', default='')
    ap.add_argument('--field', default='roofline.kernel_us_per_launch', help='dotted path of the number to compare (us; lower is better)')
    ap.add_argument('--abi', default='', help='arm=version[,arm=version]: arms built for an earlier, layout-compatible ABI version')
    ap.add_argument('--tree', default='', help='arm=path[,arm=path]: an exported tree of an earlier round (git archive + its library)')
    ap.add_argument('bench_args', nargs='*')
    a = ap.parse_args()
    vdir = os.path.join(ROOT, 'safe_control_gym_amd', 'lib', 'variants')
    arms = []
    for f in sorted(os.listdir(vdir)):
        if f.endswith('.so'):
            arms.append((f[:-3], os.path.join(vdir, f), {}))
    if a.arms:
        keep = set(a.arms.split(','))
        arms = [x for x in arms if x[0] in keep]
    abis = dict(kv.split('=') for kv in a.abi.split(',') if kv)
    # whole source trees of earlier rounds (their own package, bench.py and library): arm `name` runs <path>/bench.py
    trees = dict(kv.split('=') for kv in a.tree.split(',') if kv)
    for name, path in trees.items():
        if not a.arms or name in a.arms.split(','):
            arms.append((name, None, {}))
    res = {name: [] for name, _, _ in arms}
    for r in range(a.rounds):
        for name, lib, envx in arms:
            j = run(lib, envx, a.steps, a.bench_args, abi=abis.get(name), tree=os.path.join(ROOT, trees[name]) if name in trees else None)
            us = j
            for part in a.field.split('.'):
                us = us.get(part) if isinstance(us, dict) else None
            res[name].append(us)
            print('round %d %-20s %s' % (r, name, us if us else j), flush=True)
    print('---- summary of %s (us: lower is better; *_value: higher is better)' % a.field)
    for name, v in sorted(res.items(), key=lambda kv: min([x for x in kv[1] if x] or [9e9])):
        vv = [x for x in v if x]
        print('%-20s min %.4g  max %.4g  all %s' % (name, min(vv) if vv else float('nan'), max(vv) if vv else float('nan'), ['%.4g' % x for x in vv]))


if __name__ == '__main__':
    main()
