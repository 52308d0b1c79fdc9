"""Oracle: the reference's serial vector-env loop and episode statistics (float64).

Test infrastructure -- see ``oracle/__init__.py``.

Reference anchors (relative to /root/reference/safe_control_gym/envs/env_wrappers):
  * vectorized_env/dummy_vec_env.py:34-59    step_wait: for i in envs: step; on done stash
                                             terminal obs/info, reset, return the reset obs
  * record_episode_statistics.py:142-169     per-env episode_return / episode_length
                                             accumulators, flushed to queues on done
"""
from collections import deque

import numpy as np


class RefVecEnv:
    def __init__(self, envs):
        self.envs = envs
        self.num_envs = len(envs)
        self.episode_return = np.zeros(self.num_envs)
        self.episode_length = np.zeros(self.num_envs)
        self.return_queue = deque()
        self.length_queue = deque()

    def reset(self):
        self.episode_return[:] = 0
        self.episode_length[:] = 0
        results = [env.reset() for env in self.envs]
        obs, infos = zip(*results)
        return np.stack(obs), {'n': infos}

    def step(self, actions):
        """``actions`` is (N, nu).  Returns obs (N, obs_dim), rew (N,), done (N,), {'n': infos}."""
        results = []
        for i in range(self.num_envs):
            obs, rew, done, info = self.envs[i].step(actions[i])
            if done:
                end_obs, end_info = np.array(obs, copy=True), dict(info)
                obs, info = self.envs[i].reset()
                info = dict(info)
                info['terminal_observation'] = end_obs
                info['terminal_info'] = end_info
            results.append([obs, rew, done, info])
        obs, rews, dones, infos = zip(*results)
        rews, dones = np.array(rews), np.array(dones)
        for i, (r, d) in enumerate(zip(rews, dones)):           # record_episode_statistics.py:144-160
            self.episode_return[i] += r
            self.episode_length[i] += 1
            if d:
                infos[i]['episode'] = {'r': self.episode_return[i], 'l': self.episode_length[i]}
                self.return_queue.append(float(self.episode_return[i]))
                self.length_queue.append(float(self.episode_length[i]))
                self.episode_return[i] = 0
                self.episode_length[i] = 0
        return np.stack(obs), rews, dones, {'n': infos}
