"""The scripts under examples/ run and print what their docstrings promise (small sizes; one child process at a time)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'examples', script)] + [str(a) for a in args], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_ppo_rollout_example(gpu):
    out = _run('ppo_rollout.py', 4096, 8)
    assert 'env-steps/s through policy + env + normalisers + buffer + GAE' in out and 'advantage mean' in out
    for mode in ('pushed', 'graph'):                   # the reference's push scheme; the rollout recorded in a HIP graph and replayed
        out = _run('ppo_rollout.py', 4096, 8, mode)
        assert 'env-steps/s through policy + env + normalisers + buffer + GAE' in out and 'nan' not in out.lower()


def test_linearize_along_trajectory_example(gpu):
    out = _run('linearize_along_trajectory.py', 4)
    assert '1200 operating points' in out
    gap = float(out.split('the reference:')[1].split()[0])
    assert gap < 1e-3                                   # a linear model one step from its expansion point
