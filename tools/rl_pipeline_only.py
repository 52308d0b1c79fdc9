#!/usr/bin/env python
"""bench.py's `rl_pipeline` (and optionally `step_api`) sub-record alone: python tools/rl_pipeline_only.py [envs] [T] [step_api]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
args = bench.parse_args([])
out = {'rl_pipeline': bench.rl_pipeline_record(args, n, T)}
if 'step_api' in sys.argv[3:]:
    out['step_api'] = bench.step_api_record(args, n, 300)
print(json.dumps(out))
