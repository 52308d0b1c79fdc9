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


def test_collect_trajectories_example(gpu):
    for fused in (0, 1):                                # per-step launches / the one-launch form
        out = _run('collect_trajectories.py', 4096, 130, fused)
        assert 'env-steps/s' in out and 'dataset: obs (130, 4096, 6) reward (130, 4096) done (130, 4096) constraint_values (130, 4096, 6)' in out
        assert int(out.split('episodes finished: ')[1].split(';')[0]) >= 4096          # 2-s episodes: every env finished one
