"""``make_vec_envs`` -- the reference's entry point for parallel rollouts, served by ONE batched env.

The reference builds ``batch_size`` Python env objects and steps them in a ``for`` loop
(``DummyVecEnv``, dummy_vec_env.py:34-49) or in worker processes over pipes (``SubprocVecEnv``,
subproc_vec_env.py:17-262); see env_wrappers/vectorized_env/__init__.py:40-65.  Here the batch is
the kernel's grid: ``make_vec_envs(env_func, batch_size=N)`` calls ``env_func(num_envs=N, ...)`` once
and wraps the result in the ``VecEnv`` protocol (vec_env.py:16-162).  ``n_processes`` is accepted and
ignored (there is nothing to fork); per-env seeds ``seed + rank`` (:28-33) become the Philox stream
index ``env_index_base + rank``.

``num_pipelines=P`` (new) is what ``n_processes`` was for: the batch is cut into P contiguous shards (the reference's
``np.array_split`` over workers, subproc_vec_env.py:28) that advance on P launch queues, ``step_async(actions, pipeline=p)`` /
``step_wait(pipeline=p)`` per shard (``PipelinedDeviceVecEnv``); results equal the unsplit batch bit for bit.
"""
from safe_control_gym_amd.envs.env_wrappers.vectorized_env.vec_env import (VecEnv, VecEnvWrapper, DeviceVecEnv,
                                                                            PipelinedDeviceVecEnv)


def pipeline_bounds(batch_size, num_pipelines):
    """Shard boundaries: ``np.array_split`` sizes, moved to multiples of 256 envs (whole workgroups) when the batch allows."""
    p = max(1, min(int(num_pipelines), int(batch_size)))
    cuts = [(batch_size * k) // p for k in range(p + 1)]
    if batch_size >= 512 * p:
        cuts = [0] + [c // 256 * 256 for c in cuts[1:-1]] + [batch_size]
    return [(cuts[k], cuts[k + 1]) for k in range(p)]


def make_vec_envs(env_func, env_configs=None, batch_size=1, n_processes=1, seed=None, num_pipelines=1):
    if env_configs is not None:
        first = env_configs[0]
        if any(cfg != first for cfg in env_configs):
            raise ValueError('[ERROR] the batched engine steps one configuration per batch; env_configs differ.')
        cfg = dict(first)
    else:
        cfg = {}
    if seed is not None:
        cfg['seed'] = seed
    if num_pipelines is None or int(num_pipelines) <= 1 or batch_size < 2:
        return DeviceVecEnv(env_func(num_envs=batch_size, **cfg))
    base = int(cfg.pop('env_index_base', 0))
    return PipelinedDeviceVecEnv([env_func(num_envs=hi - lo, env_index_base=base + lo, **cfg)
                                  for lo, hi in pipeline_bounds(batch_size, num_pipelines)])
