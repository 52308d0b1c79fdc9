"""``make_vec_envs`` -- the reference's entry point for parallel rollouts, served by ONE batched env.

The reference builds ``batch_size`` Python env objects and steps them in a ``for`` loop
(``DummyVecEnv``, dummy_vec_env.py:34-49) or in worker processes over pipes (``SubprocVecEnv``,
subproc_vec_env.py:17-262); see env_wrappers/vectorized_env/__init__.py:40-65.  Here the batch is
the kernel's grid: ``make_vec_envs(env_func, batch_size=N)`` calls ``env_func(num_envs=N, ...)`` once
and wraps the result in the ``VecEnv`` protocol (vec_env.py:16-162).  ``n_processes`` is accepted and
ignored (there is nothing to fork); per-env seeds ``seed + rank`` (:28-33) become the Philox stream
index ``env_index_base + rank``.
"""
from safe_control_gym_amd.envs.env_wrappers.vectorized_env.vec_env import VecEnv, VecEnvWrapper, DeviceVecEnv


def make_vec_envs(env_func, env_configs=None, batch_size=1, n_processes=1, seed=None):
    if env_configs is not None:
        first = env_configs[0]
        if any(cfg != first for cfg in env_configs):
            raise ValueError('[ERROR] the batched engine steps one configuration per batch; env_configs differ.')
        cfg = dict(first)
    else:
        cfg = {}
    if seed is not None:
        cfg['seed'] = seed
    env = env_func(num_envs=batch_size, **cfg)
    return DeviceVecEnv(env)
