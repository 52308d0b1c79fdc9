"""Episode statistics for the batched env (reference: env_wrappers/record_episode_statistics.py:92-169).

The per-env accumulators (``episode_return[i] += r``, ``episode_length[i] += 1``, flush on done,
:144-160) are fused into the step kernel (``ep_return`` / ``ep_length`` planes; the finished episode lands in the
env's ``terminal_record`` row);
this wrapper exposes them under the reference's names and keeps the ``return_queue`` /
``length_queue`` deques by copying ONLY the finished episodes to the host when asked.
"""
from collections import deque

import torch

from safe_control_gym_amd.envs.env_wrappers.vectorized_env.vec_env import VecEnvWrapper


class VecRecordEpisodeStatistics(VecEnvWrapper):
    def __init__(self, venv, deque_size=None, **kwargs):
        super().__init__(venv)
        self.deque_size = deque_size
        self.return_queue = deque(maxlen=deque_size)
        self.length_queue = deque(maxlen=deque_size)
        self.episode_stats, self.accumulated_stats, self.queued_stats = {}, {}, {}

    @property
    def episode_return(self):
        return self.venv.unwrapped.env.episode_return

    @property
    def episode_length(self):
        return self.venv.unwrapped.env.episode_length

    def add_tracker(self, name, init_value, mode='accumulate'):
        """Track ``info[name]`` (an [N] tensor) per episode (record_episode_statistics.py:115-130)."""
        env = self.venv.unwrapped.env
        self.episode_stats[name] = torch.full((env.num_envs,), float(init_value), device=env.device)
        if mode == 'accumulate':
            self.accumulated_stats[name] = init_value
        elif mode == 'queue':
            self.queued_stats[name] = deque(maxlen=self.deque_size)
        else:
            raise Exception('Tracker mode not implemented.')

    def reset(self, **kwargs):
        for key in self.episode_stats:
            self.episode_stats[key].zero_()
        return self.venv.reset(**kwargs)

    def step_wait(self):
        obs, reward, done, info = self.venv.step_wait()
        for key, acc in self.episode_stats.items():
            if key in info:
                acc += info[key].to(acc.dtype)
        if 'episode' in info:
            idx = torch.nonzero(done).flatten()
            if idx.numel():
                # a bounded deque keeps the LAST deque_size entries of what it is extended with: only those cross to the host (at 2^20
                # envs tens of thousands of episodes end per step; a queue of 100 needs 100 of them)
                tail = idx if self.deque_size is None or idx.numel() <= self.deque_size else idx[-self.deque_size:]
                self.return_queue.extend(info['episode']['r'][tail].tolist())
                self.length_queue.extend(info['episode']['l'][tail].tolist())
                for key, acc in self.episode_stats.items():
                    if key in self.accumulated_stats:
                        self.accumulated_stats[key] += float(acc[idx].sum())
                    if key in self.queued_stats:
                        self.queued_stats[key].extend(acc[tail].tolist())
                    acc[idx] = 0
        return obs, reward, done, info


# The reference also has the single-env wrapper of this name (record_episode_statistics.py:11-89); a batched env is served by
# the one class above whatever its size.
RecordEpisodeStatistics = VecRecordEpisodeStatistics
