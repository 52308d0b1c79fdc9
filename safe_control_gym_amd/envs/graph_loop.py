"""Recording a caller's step loop -- policy kernels, ``env.step`` / ``step_async`` / ``step_wait`` / ``rollout`` -- in ONE HIP graph.

Why: from Python a step costs 5-13 us of host time (two to three times that with a torch policy in the loop), which is what
bounds every batch below ~2^17 envs and every pipelined loop (``make_vec_envs(num_pipelines=P)``).  A graph replays K steps with
one launch.  ``torch.cuda.graph`` alone records the loop (the library's launches and event fork / joins are capturable), but a
replayed graph repeats its nodes' parameters -- among them the step serial, the counter word of every Philox draw (resets,
disturbance noise): each replay would redraw the recording's numbers.  ``capture_loop`` records the steps with serial OFFSETS and
a device word (``scg_step_io.serial_word``) that it rewrites before every replay, so a replayed loop equals eager stepping bit
for bit (tests/test_pipelines_gpu.py::test_captured_loop_replays_equal_eager_stepping).

The reference has nothing to mirror here (its loops are Python over NumPy envs: controllers/ppo/ppo.py:247-293); this is what its
``for`` loop over ``rollout_steps`` becomes when the env, the policy and the buffer live on one device.
"""
import gc

import torch


class CapturedLoop:
    """``replay()`` runs the recorded stretch once more on the current stream; ``steps`` = iterations of ``body`` per replay."""

    def __init__(self, envs, body, steps, warmup=2):
        self.envs, self.steps = list(envs), int(steps)
        if self.steps < 1:
            raise ValueError('[ERROR] capture_loop(): steps must be >= 1.')
        for k in range(warmup):                      # eager: lazy initialisation (queues, events, torch's kernels and workspaces)
            body(k)
        torch.cuda.synchronize()
        from safe_control_gym_amd.math_and_models.normalization import live_trackers
        trackers = [(r, r._cur) for r in live_trackers()]
        begun = []
        try:
            for e in self.envs:
                e._begin_capture()
                begun.append(e)
        except Exception:
            for e in begun:                          # an env that refused (fresh outputs, an adversary): the others leave recording mode
                e._end_capture()
            raise
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        # Nothing but the recorded work may touch the runtime while the capture is open: a cyclic-garbage collection that happens to
        # run inside it would finalise whatever env objects are waiting (scg_destroy: hipFree, stream / event synchronisation) and
        # invalidate the capture.  Collect now, keep the collector off until the capture has ended; other threads are not our business
        # (capture_error_mode='thread_local').
        gc.collect()
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.stream(side):
                with torch.cuda.graph(self.graph, stream=side, capture_error_mode='thread_local'):
                    for k in range(self.steps):
                        body(k)
        finally:
            if was_enabled:
                gc.enable()
            self.serials = [e._end_capture() for e in self.envs]
        torch.cuda.current_stream().wait_stream(side)
        # host-side state that alternates per call must have come round: a replay runs no Python, so the nodes of the NEXT replay start
        # from the statistics record the recording started from
        odd = [r for r, cur in trackers if r._cur != cur]
        if odd:
            for r in odd:
                r._cur = 1 - r._cur                  # nothing ran: back to the record the recording started from
            self.graph = None
            raise ValueError('[ERROR] capture_loop(): %d running-statistics tracker(s) were updated an odd number of times in the recorded '
                             'stretch; their records are double-buffered, so a replayable stretch holds an even number of updating normaliser '
                             'calls (an even number of steps with one call per step).' % len(odd))

    def replay(self):
        for e in self.envs:
            e._before_replay()                       # this replay's base serial -> the env's device word (a 4-byte fill on this stream)
        self.graph.replay()
        for e, used in zip(self.envs, self.serials):
            e._serial += used


def capture_loop(envs, body, steps, warmup=2):
    """Record ``steps`` iterations of ``body(k)`` -- capture-safe torch code around ``env.step`` (no host synchronisation, no
    data-dependent shapes) -- for the batched env(s) ``envs`` (one env, or the shards of a ``PipelinedDeviceVecEnv``) and return a
    ``CapturedLoop``.  The envs must step in place (``copy_outputs=False`` / ``info_level='minimal'``): their output views are the
    same tensors in every step, which is what lets a policy recorded once read the right observation on every replay.  ``body``
    runs ``warmup`` times eagerly first (those steps count: the envs advance).

    Host-side state that alternates per call must come round per replay: the normalisers double-buffer their statistics record
    (``RunningMeanStd._records``), so a recorded stretch has to contain an EVEN number of updating normaliser calls (``steps`` even
    with one call per step) -- the nodes of the next replay start from the record the recording started from."""
    if not isinstance(envs, (list, tuple)):
        envs = getattr(envs, 'envs', [envs])
    return CapturedLoop(envs, body, steps, warmup)
