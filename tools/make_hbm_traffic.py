#!/usr/bin/env python3
"""profiles/hbm_traffic.json from the PMC summaries of a measurement session (profiles/pmc.sh).

HBM bytes per step = (2 x FETCH_SIZE + WRITE_SIZE) per launch x launches per step: FETCH_SIZE / WRITE_SIZE are KiB per
launch from separate rocprofv3 --pmc passes; FETCH_SIZE is doubled per the gfx950 correction (MI355X_MICROARCH.md, HBM:
128-B requests tallied at 64 B).  usage: make_hbm_traffic.py profiles/r02 'round 2'"""
import json
import re
import sys


def read(path):
    vals = {}
    for line in open(path):
        m = re.match(r'(\w+)\s+per-launch avg ([0-9.e+]+)\s+\(launches (\d+)\)', line)
        if m:
            vals[m.group(1)] = (float(m.group(2)), int(m.group(3)))
    return vals


def entry(vals, moved, survey, source, collected, launches_per_step=2.0, envs=1 << 20):
    f, w = vals['FETCH_SIZE'][0], vals['WRITE_SIZE'][0]
    per_step = (2.0 * f + w) * 1024.0 * launches_per_step
    return {'bytes_per_launch': per_step, 'fetch_size_kib': f * launches_per_step, 'write_size_kib': w * launches_per_step, 'envs': envs,
            'launches_per_step': launches_per_step, 'collected': collected, 'moved_bytes': moved * envs, 'survey_bytes': survey * envs,
            'traffic_over_moved': per_step / (moved * envs),
            'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/pmc.sh), per-launch mean over %d launches '
                    'of scg_step_kernel (%s); FETCH_SIZE doubled per the gfx950 correction; scg_rollout issues 2 concurrent half-batch '
                    'launches per step: the figures are per STEP of the whole batch' % (vals['FETCH_SIZE'][1], source)}


if __name__ == '__main__':
    d, collected = sys.argv[1], sys.argv[2]
    out = {'c5': entry(read(d + '/pmc_summary.txt'), 133, 141, d + '/pmc_summary.txt', collected),
           'c4': entry(read(d + '/pmc_summary_bare.txt'), 125, 125, d + '/pmc_summary_bare.txt', collected)}
    json.dump(out, open('profiles/hbm_traffic.json', 'w'), indent=1)
    for k, v in out.items():
        print(k, '%.2f MB per step, traffic / moved = %.3f' % (v['bytes_per_launch'] / 1e6, v['traffic_over_moved']))
